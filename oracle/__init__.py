"""CPU oracle for the tensor-truth retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import from here, and only as the
checker -- never as the thing measured or shipped.  The product path
(``tensor_truth_amd``) fails loudly when the HIP library is missing; it never
falls back to this code.

Parity status: **parity unpinned by the reference's own tests** (SURVEY.md §4,
§8c): the reference mocks every embedder / vector store / reranker call and
holds no golden vectors for this path.  The oracle is therefore pinned the
only way available in the build container:

* encoder / pooling / classification head: checked against the importable
  third-party implementation the reference would run (transformers
  ``XLMRobertaModel``, ``XLMRobertaForSequenceClassification``, ``BertModel``)
  on seeded synthetic weights -- ``tests/golden/make_golden.py`` generates the
  fixtures, ``tests/test_oracle_golden.py`` checks them on every run;
* host logic (retrieval metrics): checked against outputs of the reference's
  own ``services/retrieval_metrics.py`` imported by file path (fixtures in
  ``tests/golden/metrics_golden.json``) and the known answers its unit tests
  hold (``tests/unit/services/test_retrieval_metrics.py:80-177``);
* scan / top-k: numpy restatement cross-checked against an independent plain-C
  restatement (``oracle/scan_ref.c``).
"""
