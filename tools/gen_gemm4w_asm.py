#!/usr/bin/env python3
"""Generates tools/gemm4w_asm.h: the hand-scheduled instruction stream of the 4-wave x 128 x 128 GEMM experiment
(tools/gemm4w_bench.cpp, variants 2 / 3) as ONE inline-asm block with fixed registers.

hipcc cannot be talked into this kernel (the C++ form of the same structure keeps half of the accumulators out of the
AGPRs and spills: 4x slower than the product kernel), so the experiment's stream is written out explicitly:

  registers   a[0:255]      accumulators, acc[m][n] = a[(8 m + n) * 4 ..], never leave the AGPRs before the epilogue
              v[0:64 S)     S staging sets of one 64-deep K-tile each: A piece j = v[64 s + 4 j ..], W piece j = v[64 s + 32 + 4 j ..]
              WF = 64 S     eight W fragments (one k-step), AF = WF + 32: two A fragments (double buffer)
  LDS         two stages of one K-tile [A 256 rows x 128 B | W 256 rows x 128 B], 16-byte chunks XOR-swizzled on (row >> 1) & 7
  per K-tile  two sub-iterations (k-steps of 32) of 64 MFMAs each, ONE barrier per sub-iteration:
                ks = 0: MFMAs of tile T + the 16 ds_writes of tile T + 1 (stage T + 1) + 8 global loads (A pieces) of tile T + S
                ks = 1: MFMAs + 8 global loads (W pieces) of tile T + S; its last group prefetches tile T + 1's first fragments
              fragment reads are issued one group (8 MFMAs) ahead; the W fragments of the next k-step replace the current
              ones one by one right behind their last use
  waits       counted: the generator tracks the in-order LDS / vector-memory queues and emits s_waitcnt lgkmcnt(n) /
              vmcnt(n) for exactly the operation an instruction needs

S = 2: a tile is loaded one tile-time (~1 us) before it is written to LDS, S = 3: two tile-times.  The unrolled body covers
lcm(2, S) tiles; for S = 3 the loop is entered at body position P0 = 2 so that K / 64 = 16 or 64 tiles (both = 4 mod 6) end
exactly at the end of the body.
"""
import sys


class Gen:
    def __init__(self, S, ablate=()):
        self.S = S
        self.ablate = set(ablate)       # timing experiments (wrong results): "nowrite", "nogload", "nobarrier", "noread", "nowait"
        self.WF = 64 * S                # two sets of eight W fragments (one per k-step parity)
        self.AF = self.WF + 64          # four A fragment buffers (read three groups ahead)
        self.TMP = self.AF + 16         # scratch VGPRs: TMP (A k-offset), TMP+1 (W k-offset)
        self.lines = []
        self.in_loop = False
        self.lds_q = []                 # in-order LDS queue: tags of issued, possibly incomplete ops
        self.vm_q = []                  # in-order vector-memory queue

    def emit(self, s):
        self.lines.append(s)

    # ---- queues ------------------------------------------------------------------------------------------
    def lds_issue(self, tag):
        self.lds_q.append(tag)
        if self.ablate and len(self.lds_q) > 12:
            del self.lds_q[:-12]

    def lds_need(self, tag):
        """all LDS ops up to and including `tag` must be complete"""
        if tag in self.lds_q:
            n_after = len(self.lds_q) - 1 - self.lds_q.index(tag)
            assert n_after <= 15, n_after
            self.emit(f"s_waitcnt lgkmcnt({n_after})")
            del self.lds_q[: len(self.lds_q) - n_after]

    def lds_drain(self):
        if self.lds_q:
            self.emit("s_waitcnt lgkmcnt(0)")
            self.lds_q = []

    def vm_issue(self, tag):
        self.vm_q.append(tag)
        if self.ablate and len(self.vm_q) > 32:
            del self.vm_q[:-32]

    def vm_need(self, tag):
        if tag in self.vm_q:
            n_after = len(self.vm_q) - 1 - self.vm_q.index(tag)
            assert n_after <= 63, n_after
            self.emit(f"s_waitcnt vmcnt({n_after})")
            del self.vm_q[: len(self.vm_q) - n_after]

    # ---- pieces ------------------------------------------------------------------------------------------
    def gload(self, tile_tag, st, op, j):
        """piece j (8 rows x 128 B) of operand op (0 A, 1 W) of the tile whose k offset is in TMP / TMP+1 -> staging set st"""
        dst = 64 * st + 32 * op + 4 * j
        base = 40 + 16 * op + 2 * j     # s[40:55] A row-piece bases, s[56:71] W
        if "nogload" in self.ablate and self.in_loop:
            return
        self.emit(f"global_load_dwordx4 v[{dst}:{dst + 3}], v{self.TMP + op}, s[{base}:{base + 1}]")
        self.vm_issue((tile_tag, op, j))

    def swrite(self, tile_tag, st, stage, op, j):
        src = 64 * st + 32 * op + 4 * j
        if "nowrite" in self.ablate and self.in_loop:
            return
        self.vm_need((tile_tag, op, j))
        addr = f"%[wr{stage}{j & 1}]"
        self.emit(f"ds_write_b128 {addr}, v[{src}:{src + 3}] offset:{32768 * op + 1024 * j}")
        self.lds_issue(("w", tile_tag, op, j))

    def read_w(self, tag, stage, ks, n):
        r = self.WF + 32 * ks + 4 * n
        if "noread" in self.ablate and self.in_loop:
            return
        self.emit(f"ds_read_b128 v[{r}:{r + 3}], %[rd{stage}{ks}w] offset:{2048 * n}")
        self.lds_issue(("wf", tag, n))

    def read_a(self, tag, stage, ks, m, buf):
        if "noread" in self.ablate and self.in_loop:
            return
        self.emit(f"ds_read_b128 v[{self.AF + 4 * buf}:{self.AF + 4 * buf + 3}], %[rd{stage}{ks}a] offset:{2048 * m}")
        self.lds_issue(("af", tag, m))

    def mfma(self, m, n, buf, ks):
        a = (8 * m + n) * 4
        w = self.WF + 32 * ks + 4 * n
        self.emit(f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{w}:{w + 3}], "
                  f"v[{self.AF + 4 * buf}:{self.AF + 4 * buf + 3}], a[{a}:{a + 3}]")

    def koff(self, label_tile_reg):
        """TMP = vaoff + koff, TMP+1 = vwoff + koff for the tile index in s34 (clamped to the last tile)"""
        self.emit("s_min_u32 s35, s34, %[last]")
        self.emit("s_lshl_b32 s35, s35, 7")
        self.emit(f"v_add_u32 v{self.TMP}, s35, %[vaoff]")
        self.emit(f"v_add_u32 v{self.TMP + 1}, s35, %[vwoff]")

    # ---- one K-tile at body position pos (virtual tile V = pos mod body) ---------------------------------------
    PD = 3          # A fragments are read PD groups (of 8 MFMAs) ahead, into buffer (group index) % 4

    def tile(self, pos, dry=False):
        """dry: emit only the reads this tile issues FOR THE NEXT TILE (the prologue uses it to leave the LDS queue exactly as a
        real tile leaves it)"""
        S = self.S
        stage, st_next = pos % 2, (pos + 1) % S          # this tile's LDS stage; staging set of tile V + 1
        st_load = pos % S                                # tile V + S reuses this tile's staging set
        V, nxt, far = ("t", pos), ("t", pos + 1), ("t", pos + S)
        if not dry:
            self.emit(f"s_add_u32 s34, s33, {S}")        # s33 = this tile's index: the tile to load is s33 + S (clamped)
            self.koff(None)
        writes = [(0, j) for j in range(8)] + [(1, j) for j in range(8)]
        for G in range(16):
            ks, m = G // 8, G % 8
            # A fragment of group G + PD (possibly of the next tile: only after this tile's ks = 0 barrier, G + PD >= 16 => G >= 13)
            Gf = G + self.PD
            if Gf < 16:
                if not dry:
                    self.read_a((V, Gf // 8), stage, Gf // 8, Gf % 8, Gf % 4)
            else:
                self.read_a((nxt, 0), stage ^ 1, 0, Gf - 16, Gf % 4)
            fill = []
            if not dry:
                self.lds_need(("af", (V, ks), m))
                fill.append(lambda ks=ks, m=m: self.gload(far, st_load, ks, m))
                if ks == 0 and m < 6:          # tile V + 1 -> the other LDS stage, done two groups before the barrier
                    for _ in range(3):
                        if writes:
                            op, jj = writes.pop(0)
                            fill.append(lambda op=op, jj=jj: self.swrite(nxt, st_next, stage ^ 1, op, jj))
            if m >= 4:                         # W fragments of the next k-step into the other set, two per group
                for n in (2 * (m - 4), 2 * (m - 4) + 1):
                    if ks == 0:
                        if not dry:
                            fill.append(lambda n=n: self.read_w((V, 1), stage, 1, n))
                    else:
                        fill.append(lambda n=n: self.read_w((nxt, 0), stage ^ 1, 0, n))
            for n in range(8):
                if not dry:
                    if m == 0:
                        self.lds_need(("wf", (V, ks), n))
                    self.mfma(m, n, G % 4, ks)
                if fill:
                    fill.pop(0)()
            while fill:
                fill.pop(0)()
            if m == 7 and not dry:
                # ks = 0: publish this wave's ds_writes of tile V + 1 (a counted wait for the LAST write: the fragment reads
                # behind it stay in flight); ks = 1: orders every wave's reads of this stage before the next tile's writes
                if ks == 0:
                    self.lds_need(("w", nxt, 1, 7))
                if not ("nobarrier" in self.ablate and self.in_loop):
                    self.emit("s_barrier")
        if not dry:
            self.emit("s_add_u32 s33, s33, 1")

    def retag(self, shift):
        """after a tile: tags ('t', pos + k) become ('t', pos + k - shift)?  Not needed: tags use absolute body positions and the
        body is generated linearly; the loop back-edge re-enters with the SAME relative queue state as the prologue leaves."""

    def build(self, p0):
        S = self.S
        body = 2 if S == 2 else 6
        e = self.emit
        # ---- setup: row-piece bases s[40:55] (A), s[56:71] (W)
        e("s_mov_b32 s33, 0")
        for op, (ptr, step) in enumerate((("%[pa]", "%[stepa]"), ("%[pw]", "%[stepw]"))):
            b = 40 + 16 * op
            e(f"s_mov_b64 s[{b}:{b + 1}], {ptr}")
            for j in range(1, 8):
                e(f"s_add_u32 s{b + 2 * j}, s{b + 2 * j - 2}, {step}")
                e(f"s_addc_u32 s{b + 2 * j + 1}, s{b + 2 * j - 1}, 0")
        for i in range(0, 256, 1):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        # ---- prologue: tiles 0 .. S-1 -> staging sets (p0 + T) % S; tile 0 -> LDS stage p0 % 2; first fragments
        for T in range(S):
            e(f"s_mov_b32 s34, {T}")
            self.koff(None)
            for op in range(2):
                for j in range(8):
                    self.gload(("t", p0 + T), (p0 + T) % S, op, j)
        for op in range(2):
            for j in range(8):
                self.swrite(("t", p0), p0 % S, p0 % 2, op, j)
        self.lds_drain()
        e("s_barrier")
        self.tile(p0 - 1, dry=True)      # the fragment reads a real tile p0 - 1 would have issued for tile p0, in its order
        entry_state = (list(self.lds_q), list(self.vm_q))
        if p0:
            e(f"s_branch L_pos{p0}_%=")
        # ---- loop body
        # queue state at the loop head must equal the state at the back edge: generate the body twice, check the fixed point
        e("L_loop_%=:")
        self.in_loop = True
        # for p0 != 0 the first pass enters in the middle: simulate the queue from there
        states = {}
        saved_lines = self.lines
        # pass 1 (from p0 to the end of the body) establishes the back-edge state; pass 2 generates the real body from pos 0
        self.lines = []
        for pos in range(p0, body):
            self.tile(pos)
        back = self.normalise(body)
        self.lines = saved_lines
        self.lds_q, self.vm_q = list(back[0]), list(back[1])
        for pos in range(body):
            e(f"L_pos{pos}_%=:")
            if pos == p0 and p0:
                states["entry"] = (list(self.lds_q), list(self.vm_q))
            self.tile(pos)
        back2 = self.normalise(body)
        assert self.ablate or back2 == back, "loop-carried queue state is not a fixed point"
        if p0:
            want = self.shift_state(entry_state, 0)
            assert states["entry"] == want, (states["entry"], want)
        self.in_loop = False
        e("s_cmp_lt_u32 s33, %[nk]")
        e("s_cbranch_scc1 L_loop_%=")
        self.lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lines.append("s_nop 15")
        self.lines.append("s_nop 15")
        # ---- epilogue: bias + bf16, 8-byte stores; bias float4 of column tile n in v[4 n ..] (staging registers are free)
        for n in range(8):
            e(f"global_load_dwordx4 v[{4 * n}:{4 * n + 3}], %[vbias], %[pbias] offset:{64 * n}")
        e("s_waitcnt vmcnt(0)")
        e(f"v_mov_b32 v{self.TMP}, %[vcoff]")
        for m in range(8):
            for n in range(8):
                a = (8 * m + n) * 4
                t = 64 + 8 * (n & 3)          # rotate over four temp groups so stores overlap conversions
                for r in range(4):
                    e(f"v_accvgpr_read_b32 v{t + r}, a{a + r}")
                e("s_nop 1")
                for r in range(4):
                    e(f"v_add_f32 v{t + r}, v{t + r}, v{4 * n + r}")
                e(f"v_cvt_pk_bf16_f32 v{t + 4}, v{t}, v{t + 1}")
                e(f"v_cvt_pk_bf16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
                e(f"global_store_dwordx2 v{self.TMP}, v[{t + 4}:{t + 5}], %[pc] offset:{32 * n}")
            e(f"v_add_u32 v{self.TMP}, %[rowstep16], v{self.TMP}")
        e("s_waitcnt vmcnt(0)")
        return body

    def normalise(self, body):
        """queue state with tile tags shifted back by one body length (what the next pass over the body sees)"""
        return self.shift_state((self.lds_q, self.vm_q), body)

    @staticmethod
    def shift_state(state, body):
        def sh(tag):
            def sh_t(t):
                return ("t", t[1] - body)
            if tag[0] == "w":
                return ("w", sh_t(tag[1]), tag[2], tag[3])
            if tag[0] in ("wf", "af"):
                (t, ks) = tag[1]
                return (tag[0], (sh_t(t), ks), tag[2])
            return (sh_t(tag[0]), tag[1], tag[2])
        return ([sh(t) for t in state[0]], [sh(t) for t in state[1]])


def main():
    out = ["// generated by tools/gen_gemm4w_asm.py -- do not edit", "#pragma once"]
    variants = {"S2": (), "S2_NOWRITE": ("nowrite",), "S2_NOGLOAD": ("nogload",), "S2_NOBARRIER": ("nobarrier",),
                "S2_NOREAD": ("noread",), "S2_MFMAONLY": ("nowrite", "nogload", "nobarrier", "noread")}
    for name, abl in variants.items():
        g = Gen(2, abl)
        g.build(0)
        text = "\\n\\t".join(g.lines)
        out.append(f"// {name}: {len(g.lines)} instructions" + (f" (ablation, wrong results: {', '.join(abl)})" if abl else ""))
        out.append(f'#define GEMM4W_ASM_{name} "{text}"')
    g = Gen(2)
    clob = [f'"v{i}"' for i in range(g.TMP + 2)] + [f'"a{i}"' for i in range(256)] + \
           ['"s33"', '"s34"', '"s35"'] + [f'"s{i}"' for i in range(40, 72)] + ['"vcc"', '"scc"', '"memory"']
    out.append("#define GEMM4W_CLOBBERS_S2 " + ", ".join(clob))
    open(sys.argv[1] if len(sys.argv) > 1 else "gemm4w_asm.h", "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
