"""Pair tokenisation of one coalesced rerank batch (400 pairs x ~292 tokens, trained 250k-piece Unigram model) and of a retrieval
batch's few query strings, as a function of the Rust tokenizer's thread count (RAYON_NUM_THREADS; read once, when the pool starts).
Usage: python tools/probes/tokenizer_threads.py   (spawns one child per setting)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_text as st

    tk = st.unigram_tokenizer()
    pairs = [(st.zipf_text(10 ** 9 + i, 15, 100000), st.zipf_text(i, 124, 100000)) for i in range(400)]
    qs = [p[0] for p in pairs[:4]]
    tk.encode_pair_batch(pairs[:8], 512)
    tk.encode_batch(qs)
    best_p = best_q = best_1 = 1e9
    for _ in range(7):
        t0 = time.perf_counter(); tk.encode_pair_batch(pairs, 512); best_p = min(best_p, time.perf_counter() - t0)
        t0 = time.perf_counter(); tk.encode_batch(qs); best_q = min(best_q, time.perf_counter() - t0)
        t0 = time.perf_counter(); tk.encode_pair_batch(pairs[:50], 512); best_1 = min(best_1, time.perf_counter() - t0)
    print(f"RAYON_NUM_THREADS={os.environ.get('RAYON_NUM_THREADS', '(default)'):>9}: 400 pairs {best_p * 1e3:7.2f} ms, 50 pairs {best_1 * 1e3:6.2f} ms, "
          f"4 query strings {best_q * 1e3:6.3f} ms  (cpus {os.cpu_count()})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    elif len(sys.argv) > 1:
        pass
    else:
        for n in (None, 4, 8, 16, 32, 64, 128):
            env = dict(os.environ)
            if n is None:
                env.pop("RAYON_NUM_THREADS", None)
            else:
                env["RAYON_NUM_THREADS"] = str(n)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)


def pool_child(workers):
    """the same 400 pairs through ingest_workers.PairTokenizerPool with `workers` single-threaded processes"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_text as st
    from tensor_truth_amd import ingest_workers as iw

    os.environ["TT_PAIR_WORKERS"] = str(workers)
    tk = st.unigram_tokenizer()
    pairs = [(st.zipf_text(10 ** 9 + i, 15, 100000), st.zipf_text(i, 124, 100000)) for i in range(400)]
    pool = iw.get_pair_pool(tk)
    pool.encode(pairs, 512)
    pool.encode(pairs, 512)
    best = best50 = 1e9
    for _ in range(7):
        t0 = time.perf_counter(); pool.encode(pairs, 512); best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter(); pool.encode(pairs[:100], 512); best50 = min(best50, time.perf_counter() - t0)
    print(f"PairTokenizerPool, {workers:3d} workers x 1 thread: 400 pairs {best * 1e3:7.2f} ms, 100 pairs {best50 * 1e3:6.2f} ms", flush=True)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "pool":
    pool_child(int(sys.argv[2]))
elif __name__ == "__main__" and len(sys.argv) == 1:
    for w in (4, 8, 16, 32):
        subprocess.run([sys.executable, os.path.abspath(__file__), "pool", str(w)], check=False)
