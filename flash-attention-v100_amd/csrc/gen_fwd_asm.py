#!/usr/bin/env python3
"""Generator for the hand-scheduled gfx950 forward inner loop (fa_fwd_asm.hip).

Emits fa_fwd_asm_gen.h: one inline-asm body per io type (bf16 / fp16) for
    D = 128, no bias, no dropout, contiguous K/V  (BASELINE config 2's path)
replacing the hot loop of kernel/fused_mha_forward.cu:148-209 of the reference.

Structure (one wave per SIMD, 512 registers, 4 waves x 64 query rows = 256-row workgroup):
  * a wave owns two 32-row q-blocks (qb0, qb1) that run HALF AN ITERATION OUT OF PHASE:
        phase 1 of iteration j : MFMA  QK1(j+1), PV1(j)     VALU  softmax0(j+1)
        phase 2 of iteration j : MFMA  QK0(j+2), PV0(j+1)   VALU  softmax1(j+1)
    so inside a phase the VALU stream (one q-block's whole softmax) is independent of the
    32 MFMAs issued beside it, and a q-block's O accumulators are idle when its (rare) rescale runs;
  * K fragments (A operand of S^T = K Q^T) live in a[192:255], Q fragments in a[128:191], O^T in
    a[0:127]; a K / V fragment register is reloaded from LDS right after its last use by q-block 1
    and is consumed by q-block 0 half an iteration later;
  * K/V tiles arrive by LDS-DMA into a ring of three 32 KiB stages (issued two iterations ahead,
    counted vmcnt, one barrier per tile);
  * every instruction is placed by this script; a checker pass inserts the s_nop / s_waitcnt the
    hardware needs (MFMA result hazards, trans forwarding, LDS return order) and reports them.

Run:  python gen_fwd_asm.py  > fa_fwd_asm_gen.h        (the header is committed; build.py re-runs this
script when it is newer than the header).
"""
import sys

# ----------------------------------------------------------------------------- register map
# VGPR (v0-v15 are left to the compiler)
V_QOFF = (16, 17)        # in, D = 64: Q voffset of qb0 / qb1 (prologue only)
V_QCB, V_QROW = 16, 17   # in, D = 128: swizzled column byte of the lane's 16-byte chunk for row (lane >> 4) of a 4-row DMA piece
                         # (rows 4p + (lane >> 4): the byte ^ 64 p), and (lane >> 4) * q_row_bytes
V_ORB, V_R0 = 18, 19     # in (uniform): bytes per O row, first row of the wave's 64
V_LSEOFF = (20, 21)      # in: LSE voffset
V_DMAK, V_DMAV = 22, 23  # in: LDS-DMA source voffsets
V_KBASE = 24             # in: 8 regs, K fragment read address (stage 0)
V_VBASE = 32             # in: V fragment read address (stage 0, V region)
V_LOG = (33, 35)         # in: lo - 4g of the lane's row (qb0, qb1); 0x3fffffff for an empty row
V_WIDTH = (34, 36)       # in: hi - lo (0 for an empty row)
V_KADDR = 37             # 8 regs, current K read addresses
V_VADDR = 45
V_MRUN = (46, 49)
V_NEGM = (47, 50)
V_L = (48, 51)
V_T = 52                 # temps v52..v79
NT = 28
V_S = (80, 112)          # S^T accumulators of qb0 / qb1: 32 regs each (kb0: 16, kb1: 16)
V_P = (144, 160)         # packed P^T: 16 regs each (4 k-steps x 4)
V_VF = 176               # V^T fragments: [ks][dblk] x 4 regs = 64
V_DMAK_CUR, V_DMAV_CUR = 68, 69     # fast loop: DMA source voffsets of the tiles being fetched (advance by one tile per iteration)
V_L2 = (70, 71)          # second row-sum chain per q-block
V_THR = (73, 74)         # rescale threshold in raw score units: (m_run + 8) / c
# causal ALiBi variant (bias = beta (key - q - off) in raw score units, beta = slope / softmax_scale):
V_BK = (77, 78)          # in/out: the integer n0 + 4g - q - off of the lane's row for the NEXT tile of q-block 0 / 1
                         # (kept exact: a float tile term accumulated over 60+ tiles drifts by 1e-3 in the LSE)
V_BETA = 79              # in (uniform): beta
V_C0 = 240               # 16 regs: beta * ((r & 3) + 8 (r >> 2)), the start value of every S^T accumulator chain
V_TRI = 240              # plain variant, the same 16 registers: the triangle of an ALIGNED diagonal 32 x 32 sub-block as a lane
                         # constant, 0 where key position (r & 3) + 8 (r >> 2) + 4 g <= row l31, -inf above (see gen_mask_routine)
# AGPR
A_O = (0, 64)            # O^T accumulators per q-block: [dblk] x 16
A_Q = (128, 160)         # Q fragments per q-block: [ks] x 4
A_KF = 192               # K fragments [kb][ks] x 4

# SGPR
S_QRS, S_KRS, S_VRS, S_ORS, S_LRS = 16, 20, 24, 28, 32
S_KTILE, S_VTILE, S_K16, S_V16 = 36, 37, 38, 39
S_C = 40                 # softmax_scale * log2(e)
S_JIN, S_NMAX, S_WLO, S_WHI = 41, 42, 43, 44
S_HIMIN = (45, 47)
S_LOMAX = (46, 48)
S_W1024 = 49
# asm-owned
S_R0, S_R1, S_R2 = 50, 51, 52
S_OOB = 53
S_KSOFF, S_VSOFF = 54, 55
S_TMP = 56               # s56..s59 temps
S_N0 = 60                # tile start key for the mask routine
S_TOP = 61               # ALiBi variant: tiles run from the diagonal DOWN (the running maximum is found in the first
                         # tiles instead of climbing by 64 slope per tile and rescaling O every time): physical tile =
                         # S_TOP - loop index, S_TOP = n_max - 1 (the variant is only used with n_min = 0)
S_SUB = 62               # s[62:63] call target, s[64:65] return address
S_RET = 64
S_MASKFN = (66, 68)      # s[66:67] mask0, s[68:69] mask1
S_RESCFN = (70, 72)      # s[70:71] rescale0, s[72:73] rescale1
S_J = 74                 # iteration counter (copy of the input s41)
S_KJ = 75                # s75..s77: 1,2,3 x 16 K rows in bytes (fast-loop DMA soffsets)
S_VJ = 78                # s78..s80
S_LAST = 80
S_RC, S_FASTLO, S_FASTEND = 81, 82, 83     # in: 1 / c; fast-loop iteration range [lo, end)
S_QRB = 84               # in (D = 128): bytes per Q row
S_HIMAX = (85, 86)       # in: max over the q-block's rows of the last visible key (a 32-key sub-block that starts past it is masked
                         # for every row: the mask routine fills it instead of testing each element)
# paged K / V (block_table; pages of 64 * 2^n keys): a tile lies inside one page, its descriptor (base, extent) is rebuilt on the
# scalar unit per tile from the block-table entry, which is requested ONE ITERATION ahead (s_load) - the reference's paged
# path is the same kernel as the contiguous one (kernel/fused_mha_forward_varlen.cu:184-193)
S_KB, S_VB = 87, 89      # in: words 0, 1 of the K / V descriptor of this kv-head at page 0, row 0 (s[20:23] / s[24:27] are rebuilt per tile)
S_BT = 92                # in: s[92:93] this sequence's block-table row (an aligned pair)
S_PSH, S_PMASK = 91, 94  # in: log2(tiles per page), tiles per page - 1
S_KPAGE, S_VPAGE = 95, 96   # in: bytes between pages
S_P32 = 87               # in (plain D = 128 bodies only - the paged bodies use s87..s97): != 0: this launch item is ONE PART of a 256-row block
                         # whose key range is split over several workgroups (fa_fwd_asm.hip: forward key split of one-wave causal launches);
                         # the epilogue then leaves the part's normalised O in fp32 (the O descriptor / row bytes describe rows of 128
                         # floats in the workspace) and its own LSE - fwd_split_merge_kernel combines the parts
S_SEQK = 97              # in: keys of this sequence (rows past it read as zeros: V rows of a page's unused tail may hold anything)
S_BLKK, S_BLKV, S_BLKN = 98, 99, 100     # owned: page of the K tile fetched this iteration (j + 4), of the V tile (j + 3), of the next K tile (in flight)

# ----------------------------------------------------------------------------- head dimension (128 or 64)
HD = 128
KS = 8                   # 16-column k-steps of S^T = K Q^T
DB = 4                   # 32-row d-blocks of O^T
ROWB = 256               # bytes per K / V row
NP = 4                   # 1-KiB LDS-DMA pieces per wave and tile
LDS_STAGE = 16384        # one K (or V) tile: 64 keys
LDS_VREGION = 3 * LDS_STAGE
VQ = 1024                # bytes of one key-quad in the blocked V image ([key / 4][dblk][4][32])
EP_PITCH = 272           # epilogue: O row + 16 bytes (bank-conflict-free 8-byte writes)
EP_ROWS = 4              # rows per 1-KiB store instruction


def set_dim(d):
    """D = 64: half the k-steps and d-blocks, 128-byte rows (8 rows per DMA piece, 2 pieces per wave and tile)."""
    global HD, KS, DB, ROWB, NP, LDS_STAGE, LDS_VREGION, VQ, A_O, A_Q, EP_PITCH, EP_ROWS
    assert d in (64, 128)
    HD, KS, DB, ROWB = d, d // 16, d // 32, 2 * d
    LDS_STAGE = 64 * ROWB
    LDS_VREGION = 3 * LDS_STAGE
    NP = LDS_STAGE // 4096
    VQ = 4 * ROWB
    A_O = (0, 16 * DB)
    A_Q = (128, 128 + 4 * KS)
    EP_PITCH = ROWB + 16
    EP_ROWS = 1024 // ROWB



def vr(b, n=1):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=1):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


def sr(b, n=1):
    return f"s[{b}:{b + n - 1}]" if n > 1 else f"s{b}"


def rl(p, b, n=1):
    return [f"{p}{b + i}" for i in range(n)]


class Ins:
    __slots__ = ("txt", "kind", "rd", "wr", "cexact", "w", "aux")

    def __init__(self, txt, kind, rd=(), wr=(), cexact=False, w=1.0):
        self.txt, self.kind, self.rd, self.wr, self.cexact, self.w = txt, kind, list(rd), list(wr), cexact, w
        self.aux = None


class Gen:
    def __init__(self, dtype, alibi=False, paged=False):
        self.dtype = dtype
        self.alibi = alibi
        self.paged = paged
        assert not (alibi and paged)
        self.reverse = alibi
        self.tri = not alibi       # lane-constant triangle for aligned diagonal sub-blocks (the ALiBi variant owns v240..v255)
        self.mf = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.out = []          # final text lines
        self.stats = {"nop_states": 0, "lgkm_waits": 0}
        self.reset_state()

    # ------------------------------------------------------------------ hazard / wait tracking
    def reset_state(self, mfma_age=8):
        # conservative variant-entry state: every S / O register may have been written by an MFMA
        # `mfma_age` wait states ago; nothing in flight on the LDS queue.
        self.now = 0
        self.last = {}
        for qb in (0, 1):
            for r in rl("v", V_S[qb], 32) + rl("a", A_O[qb], 16 * DB):
                self.last[r] = (-mfma_age, "mfma", None)
        self.lds_q = []        # outstanding LDS reads: list of sets of written regs, in issue order
        self.srcc_rd = {}      # reg -> state index of the last MFMA reading it as srcC

    def _need(self, ins):
        need = 0
        for r in ins.rd:
            lw = self.last.get(r)
            if lw is None:
                continue
            t, k, tup = lw
            gap = self.now - t - 1          # wait states already between producer and consumer
            req = 0
            if k == "mfma":
                if ins.kind == "mfma" and ins.cexact and r in ins.wr:
                    req = 0                 # same accumulator back to back
                else:
                    req = 12
            elif k in ("valu", "trans"):
                if ins.kind == "mfma":
                    req = 2
                elif ins.kind == "swap":
                    req = 2
                elif k == "trans" and ins.kind in ("valu", "trans", "swap", "vmem", "dma", "lds"):
                    req = 1
                elif ins.kind in ("vmem", "dma") and r.startswith("s"):
                    req = 5
                elif ins.kind in ("vmem",) and r.startswith("v"):
                    req = 0
            elif k == "salu":
                if r == "m0" and ins.kind == "dma":
                    req = 1
            elif k == "store4":
                pass
            need = max(need, req - gap)
        for r in ins.wr:
            lw = self.last.get(r)
            if lw is not None:
                t, k, tup = lw
                gap = self.now - t - 1
                if k == "mfma" and ins.kind != "mfma":
                    need = max(need, 12 - gap)          # WAW behind an MFMA
                if k == "store4":
                    need = max(need, 2 - gap)
            if ins.kind != "mfma" and r in self.srcc_rd:
                need = max(need, 16 - (self.now - self.srcc_rd[r] - 1))   # WAR on an MFMA's srcC
        return max(need, 0)

    ko = frozenset()           # timing experiments only (results are wrong): kinds of instructions left out

    def emit(self, ins):
        if self.ko:
            k = ins.kind
            role = getattr(self, "role", "")
            if role and any(x == f"{role}_{kk}" for x in self.ko for kk in ((k,) + (("valu",) if k in ("trans", "swap") else ()))):
                return
            if (("dma" in self.ko and k == "dma") or ("lds" in self.ko and k == "lds") or ("mfma" in self.ko and k == "mfma") or
                    ("ldsk" in self.ko and k == "lds" and "b128" in ins.txt) or ("ldsv" in self.ko and k == "lds" and "tr_b16" in ins.txt) or
                    ("valu" in self.ko and k in ("valu", "trans", "swap") and ins.aux != "keep") or
                    ("salu" in self.ko and k == "salu" and "m0" not in ins.wr and ins.aux != "keep")):
                return
        if ins.kind == "label" or ins.kind == "raw":
            self.out.append(ins.txt)
            return
        # LDS return order: wait for the youngest outstanding read that produces one of our operands
        touched = set(ins.rd) | set(ins.wr)
        idx = -1
        for i, regs in enumerate(self.lds_q):
            if regs & touched:
                idx = i
        if idx >= 0:
            n_after = len(self.lds_q) - 1 - idx
            if n_after < 15:
                self.out.append(f"s_waitcnt lgkmcnt({n_after})")
                self.now += 1
                self.stats["lgkm_waits"] += 1
            self.lds_q = self.lds_q[idx + 1:]
        n = self._need(ins)
        while n > 0:
            k = min(n, 8)
            self.out.append(f"s_nop {k - 1}")
            self.now += k
            self.stats["nop_states"] += k
            n -= k
        self.out.append(ins.txt)
        kind = {"swap": "valu", "dma": "vmem"}.get(ins.kind, ins.kind)
        for r in ins.wr:
            self.last[r] = (self.now, kind, None)
        if kind == "mfma":
            for r in ins.rd:
                if ins.cexact and r in ins.wr:
                    self.srcc_rd[r] = self.now
        if kind == "lds":
            self.lds_q.append(set(ins.wr))
            if len(self.lds_q) > 15:            # the 4-bit counter stalls issue at 15 outstanding: older ones have returned
                self.lds_q = self.lds_q[-15:]
        self.now += 1

    def raw(self, txt):
        self.out.append(txt)
        self.now += 1

    def drain_lds(self):
        if self.lds_q:
            self.out.append("s_waitcnt lgkmcnt(0)")
            self.now += 1
            self.lds_q = []

    # ------------------------------------------------------------------ instruction builders
    def mfma(self, dpre, dbase, apre, abase, bpre, bbase, first):
        d = f"{dpre}[{dbase}:{dbase + 15}]"
        a = f"{apre}[{abase}:{abase + 3}]"
        b = f"{bpre}[{bbase}:{bbase + 3}]"
        c = "0" if first else d
        rd = rl(apre, abase, 4) + rl(bpre, bbase, 4) + ([] if first else rl(dpre, dbase, 16))
        return Ins(f"{self.mf} {d}, {a}, {b}, {c}", "mfma", rd, rl(dpre, dbase, 16), cexact=not first)

    def qk_mfmas(self, qb):
        """S^T[kb] (+)= K[kb][ks] Q_qb[ks]^T, ks-major so that consecutive MFMAs alternate accumulators."""
        out = []
        for ks in range(KS):
            for kb in range(2):
                m = self.mfma("v", V_S[qb] + 16 * kb, "a", A_KF + (kb * KS + ks) * 4, "a", A_Q[qb] + 4 * ks, ks == 0)
                if ks == 0 and self.alibi:      # the chain starts from beta * (key position inside the 32-key block)
                    d, a, b = V_S[qb] + 16 * kb, A_KF + (kb * KS) * 4, A_Q[qb]
                    m = Ins(f"{self.mf} {vr(d, 16)}, {ar(a, 4)}, {ar(b, 4)}, {vr(V_C0, 16)}", "mfma",
                            rl("a", a, 4) + rl("a", b, 4) + rl("v", V_C0, 16), rl("v", d, 16))
                out.append(m)
        return out

    def pv_mfmas(self, qb):
        """O^T[dblk] += V^T[dblk][ks] P_qb[ks]^T."""
        out = []
        for ks in range(4):
            for d in range(DB):
                out.append(self.mfma("a", A_O[qb] + 16 * d, "v", V_VF + (ks * DB + d) * 4,
                                     "v", V_P[qb] + 4 * ks, False))
        return out

    def k_read(self, kb, ks, fast=None):
        b = A_KF + (kb * KS + ks) * 4
        half = 32 * ROWB                       # 32 keys
        areg, off = (V_KADDR + ks, kb * half) if fast is None else (V_KBASE + ks, fast[1] + kb * half)
        return Ins(f"ds_read_b128 {ar(b, 4)}, v{areg} offset:{off}", "lds", [f"v{areg}"], rl("a", b, 4))

    def v_reads(self, d, ks, fast=None):
        b = V_VF + (ks * DB + d) * 4
        areg, off = (V_VADDR, ks * 4 * VQ + d * 256) if fast is None else (V_VBASE, fast[0] + ks * 4 * VQ + d * 256)
        assert off + 2 * VQ < 65536
        return [Ins(f"ds_read_b64_tr_b16 {vr(b, 2)}, v{areg} offset:{off}", "lds", [f"v{areg}"], rl("v", b, 2)),
                Ins(f"ds_read_b64_tr_b16 {vr(b + 2, 2)}, v{areg} offset:{off + 2 * VQ}", "lds", [f"v{areg}"],
                    rl("v", b + 2, 2))]

    def valu(self, txt, rd, wr, kind="valu", w=1.0):
        return Ins(txt, kind, rd, wr, w=w)

    def softmax(self, qb):
        """max -> (rare) rescale call -> exp2 / row sum / pack for one q-block's 64-key tile."""
        S = V_S[qb]
        P = V_P[qb]
        T = V_T
        out = []
        s = [f"v{S + i}" for i in range(32)]
        # ---- row maximum over the lane's 32 keys: max3 tree
        def max_tree(vals, tn):
            while len(vals) > 1:
                nxt = []
                i = 0
                while i + 2 < len(vals):
                    t = f"v{T + tn}"
                    tn += 1
                    out.append(self.valu(f"v_max3_f32 {t}, {vals[i]}, {vals[i + 1]}, {vals[i + 2]}",
                                         [vals[i], vals[i + 1], vals[i + 2]], [t]))
                    nxt.append(t)
                    i += 3
                rest = vals[i:]
                if len(rest) == 2 and not nxt:
                    t = f"v{T + tn}"
                    tn += 1
                    out.append(self.valu(f"v_max_f32 {t}, {rest[0]}, {rest[1]}", rest, [t]))
                    nxt.append(t)
                    rest = []
                vals = nxt + rest
            return vals[0], tn
        negms = None
        if not self.alibi:
            mx, tn = max_tree(list(s), 0)
            assert tn <= 16
        else:
            # the two 32-key blocks carry different tile terms: max per block, then + beta (n0 + 32 kb + 4g - q - off)
            dist, beta = f"v{V_BK[qb]}", f"v{V_BETA}"
            mx0, tn = max_tree(list(s[:16]), 0)
            mx1, tn = max_tree(list(s[16:]), tn)
            assert tn <= 16
            t0, bk1, t1, mx, bk = f"v{T}", f"v{T + 1}", f"v{T + 2}", f"v{T + 3}", f"v{T + 6}"
            out.append(self.valu(f"v_cvt_f32_i32 {bk}, {dist}", [dist], [bk]))
            out.append(self.valu(f"v_add_u32 {bk1}, 32, {dist}", [dist], [bk1]))
            out.append(self.valu(f"v_mul_f32 {bk}, {beta}, {bk}", [beta, bk], [bk]))
            out.append(self.valu(f"v_cvt_f32_i32 {bk1}, {bk1}", [bk1], [bk1]))
            out.append(self.valu(f"v_add_f32 {t0}, {mx0}, {bk}", [mx0, bk], [t0]))
            out.append(self.valu(f"v_mul_f32 {bk1}, {beta}, {bk1}", [beta, bk1], [bk1]))
            out.append(self.valu(f"v_add_f32 {t1}, {mx1}, {bk1}", [mx1, bk1], [t1]))
            out.append(self.valu(f"v_max_f32 {mx}, {t0}, {t1}", [t0, t1], [mx]))
            negms = (f"v{T + 4}", f"v{T + 5}", bk, bk1)
        ta, td, tl = f"v{T + 20}", f"v{T + 21}", f"v{T + 22}"
        out.append(self.valu(f"v_mov_b32 {ta}, {mx}", [mx], [ta]))
        out.append(Ins(f"v_permlane32_swap_b32 {ta}, {mx}", "swap", [ta, mx], [ta, mx]))
        mxr = f"v{T + 23 + qb}"                       # raw (unscaled) row max, read by the rescale routine
        out.append(self.valu(f"v_max_f32 {mxr}, {ta}, {mx}", [ta, mx], [mxr]))
        out.append(self.valu(f"v_cmp_lt_f32 vcc, v{V_THR[qb]}, {mxr}", [mxr, f"v{V_THR[qb]}"], ["vcc"]))
        call = Ins(f"s_cbranch_vccz L_norescale_{self.uid()}_%=", "call_rescale", ["vcc"], [], w=1.0)
        call.aux = qb
        out.append(call)
        # ---- exp2(s c - m), row sum (two chains), pack
        negm = f"v{V_NEGM[qb]}"
        negm_of = [negm, negm]
        if negms:                                      # exponent = c (S + tile term) - m: one offset per 32-key block
            n0_, n1_, bk, bk1 = negms
            out.append(self.valu(f"v_fma_f32 {n0_}, s{S_C}, {bk}, {negm}", [bk, negm], [n0_]))
            out.append(self.valu(f"v_fma_f32 {n1_}, s{S_C}, {bk1}, {negm}", [bk1, negm], [n1_]))
            negm_of = [n0_, n1_]
        l0, l1 = f"v{V_L[qb]}", f"v{V_L2[qb]}"
        # software pipeline over the 32 elements: fma(r), exp(r-1), add(r-3), pack(pair) - transcendentals never sit
        # back to back (v_exp_f32 re-issues after ~8.5 cycles, a plain VALU after ~5: probe_trans_rate)
        def pack(r):
            kb, rr = r // 16, r % 16
            ks, e = 2 * kb + rr // 8, (rr % 8) // 2
            dst = f"v{P + 4 * ks + e}"
            return self.valu(f"{self.cvt} {dst}, {s[r - 1]}, {s[r]}", [s[r - 1], s[r]], [dst])
        for r in range(32 + 3):
            if r < 32:
                nm = negm_of[r // 16]
                out.append(self.valu(f"v_fma_f32 {s[r]}, {s[r]}, s{S_C}, {nm}", [s[r], nm], [s[r]]))
            if 0 <= r - 1 < 32:
                out.append(self.valu(f"v_exp_f32 {s[r - 1]}, {s[r - 1]}", [s[r - 1]], [s[r - 1]], kind="trans", w=1.6))
            if 0 <= r - 3 < 32:
                q = r - 3
                ll = l0 if q % 2 == 0 else l1
                out.append(self.valu(f"v_add_f32 {ll}, {ll}, {s[q]}", [ll, s[q]], [ll]))
                if q % 2 == 1:
                    out.append(pack(q))
        if negms:                                      # next tile: 64 keys further
            dist = f"v{V_BK[qb]}"
            out.append(self.valu(f"v_subrev_u32 {dist}, 64, {dist}", [dist], [dist]))      # tiles run downwards
        return out

    _uid = 0

    def uid(self):
        Gen._uid += 1
        return Gen._uid

    # ------------------------------------------------------------------ DMA + bookkeeping stream
    def misc_stream(self):
        """LDS-DMA of K(j+4) -> slot R0 and V(j+3) -> slot R2, address updates for the next iteration, slot rotation.
        Returned as groups: (list of Ins, tag)."""
        g = []
        t0 = S_TMP
        offs = []
        offs.append(Ins(f"s_add_u32 s{t0}, s{S_J}, 4", "salu", [f"s{S_J}"], [f"s{t0}", "scc"], w=0.5))
        tk = t0
        if self.reverse:
            tk = t0 + 1
            offs.append(Ins(f"s_sub_u32 s{tk}, s{S_TOP}, s{t0}", "salu", [f"s{t0}"], [f"s{tk}", "scc"], w=0.5))
        offs.append(Ins(f"s_mul_i32 s{S_KSOFF}, s{tk}, s{S_KTILE}", "salu", [f"s{tk}"], [f"s{S_KSOFF}"], w=0.5))
        offs.append(Ins(f"s_cmp_lt_i32 s{t0}, s{S_NMAX}", "salu", [f"s{t0}"], ["scc"], w=0.5))
        offs.append(Ins(f"s_cselect_b32 s{S_KSOFF}, s{S_KSOFF}, s{S_OOB}", "salu", ["scc", f"s{S_KSOFF}"], [f"s{S_KSOFF}"], w=0.5))
        offs.append(Ins(f"s_add_u32 s{t0}, s{S_J}, 3", "salu", [f"s{S_J}"], [f"s{t0}", "scc"], w=0.5))
        if self.reverse:
            offs.append(Ins(f"s_sub_u32 s{tk}, s{S_TOP}, s{t0}", "salu", [f"s{t0}"], [f"s{tk}", "scc"], w=0.5))
        offs.append(Ins(f"s_mul_i32 s{S_VSOFF}, s{tk}, s{S_VTILE}", "salu", [f"s{tk}"], [f"s{S_VSOFF}"], w=0.5))
        offs.append(Ins(f"s_cmp_lt_i32 s{t0}, s{S_NMAX}", "salu", [f"s{t0}"], ["scc"], w=0.5))
        offs.append(Ins(f"s_cselect_b32 s{S_VSOFF}, s{S_VSOFF}, s{S_OOB}", "salu", ["scc", f"s{S_VSOFF}"], [f"s{S_VSOFF}"], w=0.5))
        offs.append(Ins(f"s_add_u32 s{t0 + 1}, s{S_R0}, s{S_W1024}", "salu", [], [f"s{t0 + 1}", "scc"], w=0.5))
        offs.append(Ins(f"s_add_u32 s{t0 + 2}, s{S_R2}, s{S_W1024}", "salu", [], [f"s{t0 + 2}", "scc"], w=0.5))
        g.append((offs, "offs"))
        for jj in range(NP):
            p = [Ins(f"s_add_u32 m0, s{t0 + 1}, {4096 * jj}", "salu", [f"s{t0 + 1}"], ["m0", "scc"], w=0.5),
                 Ins(f"buffer_load_dwordx4 v{V_DMAK}, {sr(S_KRS, 4)}, s{S_KSOFF} offen lds", "dma",
                     ["m0", f"v{V_DMAK}", f"s{S_KSOFF}"], [], w=4.0),
                 Ins(f"s_add_u32 s{S_KSOFF}, s{S_KSOFF}, s{S_K16}", "salu", [f"s{S_KSOFF}"], [f"s{S_KSOFF}", "scc"], w=0.5)]
            g.append((p, "dma"))
        for jj in range(NP):
            p = [Ins(f"s_add_u32 m0, s{t0 + 2}, {LDS_VREGION + 4096 * jj}", "salu", [f"s{t0 + 2}"], ["m0", "scc"], w=0.5),
                 Ins(f"buffer_load_dwordx4 v{V_DMAV}, {sr(S_VRS, 4)}, s{S_VSOFF} offen lds", "dma",
                     ["m0", f"v{V_DMAV}", f"s{S_VSOFF}"], [], w=4.0),
                 Ins(f"s_add_u32 s{S_VSOFF}, s{S_VSOFF}, s{S_V16}", "salu", [f"s{S_VSOFF}"], [f"s{S_VSOFF}", "scc"], w=0.5)]
            g.append((p, "dma"))
        return g

    # ------------------------------------------------------------------ paged K / V
    def paged_desc(self, which, tile_s, blk_s):
        """SALU: descriptor words 0..2 of the tile `tile_s` (SGPR) whose page is in `blk_s`: base = pool base + page * page
        bytes + (tile inside the page) * tile bytes, extent = the tile's rows inside the sequence (0 .. 64) in bytes.
        Temps s62..s65 (idle outside the called routines; the group is emitted in one piece)."""
        rs, base, page, tileb, s16 = ((S_KRS, S_KB, S_KPAGE, S_KTILE, S_K16) if which == "k" else (S_VRS, S_VB, S_VPAGE, S_VTILE, S_V16))
        t1, t2, t3 = S_SUB, S_SUB + 1, S_RET
        I = lambda txt, rd, wr: Ins(txt, "salu", rd, wr, w=0.5)
        return [
            I(f"s_lshl_b32 s{t1}, s{tile_s}, 6", [f"s{tile_s}"], [f"s{t1}"]),
            I(f"s_sub_i32 s{t1}, s{S_SEQK}, s{t1}", [f"s{t1}"], [f"s{t1}", "scc"]),
            I(f"s_max_i32 s{t1}, s{t1}, 0", [f"s{t1}"], [f"s{t1}", "scc"]),
            I(f"s_min_i32 s{t1}, s{t1}, 64", [f"s{t1}"], [f"s{t1}", "scc"]),
            I(f"s_mul_i32 s{rs + 2}, s{t1}, s{s16}", [f"s{t1}"], [f"s{rs + 2}"]),
            I(f"s_lshr_b32 s{rs + 2}, s{rs + 2}, 4", [f"s{rs + 2}"], [f"s{rs + 2}", "scc"]),          # rows * row bytes
            I(f"s_and_b32 s{t2}, s{tile_s}, s{S_PMASK}", [f"s{tile_s}"], [f"s{t2}", "scc"]),
            I(f"s_mul_i32 s{t2}, s{t2}, s{tileb}", [f"s{t2}"], [f"s{t2}"]),
            I(f"s_mul_hi_u32 s{t3}, s{blk_s}, s{page}", [f"s{blk_s}"], [f"s{t3}"]),
            I(f"s_mul_i32 s{t1}, s{blk_s}, s{page}", [f"s{blk_s}"], [f"s{t1}"]),
            I(f"s_add_u32 s{t1}, s{t1}, s{t2}", [f"s{t1}", f"s{t2}"], [f"s{t1}", "scc"]),
            I(f"s_addc_u32 s{t3}, s{t3}, 0", [f"s{t3}", "scc"], [f"s{t3}", "scc"]),
            I(f"s_add_u32 s{rs}, s{base}, s{t1}", [f"s{t1}"], [f"s{rs}", "scc"]),
            I(f"s_addc_u32 s{rs + 1}, s{base + 1}, s{t3}", [f"s{t3}", "scc"], [f"s{rs + 1}", "scc"]),
        ]

    def paged_request(self, tile_expr_lines, dst):
        """raw lines: dst <- block_table[min(tile, n_max - 1) >> PSH]; tile in s62 after `tile_expr_lines`."""
        t = S_SUB
        return tile_expr_lines + [
            f"s_sub_u32 s{t + 1}, s{S_NMAX}, 1",
            f"s_min_i32 s{t}, s{t}, s{t + 1}",
            f"s_max_i32 s{t}, s{t}, 0",
            f"s_lshr_b32 s{t}, s{t}, s{S_PSH}",
            f"s_lshl_b32 s{t}, s{t}, 2",
            f"s_load_dword s{dst}, {sr(S_BT, 2)}, s{t}",
        ]

    def misc_stream_paged(self, fast):
        """K(j+4) -> slot R0, V(j+3) -> slot R2 out of a paged cache: one group rotates the page registers and rebuilds the two
        descriptors, the pieces then only differ in M0 and the 16-row scalar offset.  Same stream in the generic and the fast
        copies (the fast ones take the ring slots as immediates)."""
        g = []
        t0 = S_TMP
        I = lambda txt, rd, wr: Ins(txt, "salu", rd, wr, w=0.5)
        offs = [I(f"s_add_u32 s{t0}, s{S_J}, 4", [f"s{S_J}"], [f"s{t0}", "scc"])]
        offs += self.paged_desc("k", t0, S_BLKK)
        offs.append(I(f"s_add_u32 s{t0}, s{S_J}, 3", [f"s{S_J}"], [f"s{t0}", "scc"]))
        offs += self.paged_desc("v", t0, S_BLKV)
        if fast is None:
            offs.append(I(f"s_add_u32 s{t0 + 1}, s{S_R0}, s{S_W1024}", [], [f"s{t0 + 1}", "scc"]))
            offs.append(I(f"s_add_u32 s{t0 + 2}, s{S_R2}, s{S_W1024}", [], [f"s{t0 + 2}", "scc"]))
        g.append((offs, "offs"))
        for (rs, vo, sj, reg, slot_s, slot_i) in ((S_KRS, V_DMAK, S_KJ, 0, t0 + 1, None if fast is None else fast[0]),
                                                   (S_VRS, V_DMAV, S_VJ, LDS_VREGION, t0 + 2, None if fast is None else fast[2])):
            for jj in range(NP):
                so = "0" if jj == 0 else f"s{sj + jj - 1}"
                if fast is None:
                    m0w = Ins(f"s_add_u32 m0, s{slot_s}, {reg + 4096 * jj}", "salu", [f"s{slot_s}"], ["m0", "scc"], w=0.5)
                else:
                    m0w = Ins(f"s_add_u32 m0, s{S_W1024}, {reg + slot_i + 4096 * jj}", "salu", [], ["m0", "scc"], w=0.5)
                g.append(([m0w, Ins(f"buffer_load_dwordx4 v{vo}, {sr(rs, 4)}, {so} offen lds", "dma", ["m0", f"v{vo}"], [], w=4.0)], "dma"))
        return g

    def misc_stream_fast(self, fast):
        """Fast loop: K(j+4) -> slot r0, V(j+3) -> slot r2; the source tile is in the voffset registers."""
        r0, r1, r2 = fast
        g = []
        for jj in range(NP):
            so = "0" if jj == 0 else f"s{S_KJ + jj - 1}"
            p = [Ins(f"s_add_u32 m0, s{S_W1024}, {r0 + 4096 * jj}", "salu", [], ["m0", "scc"], w=0.5),
                 Ins(f"buffer_load_dwordx4 v{V_DMAK_CUR}, {sr(S_KRS, 4)}, {so} offen lds", "dma",
                     ["m0", f"v{V_DMAK_CUR}"], [], w=4.0)]
            if jj == NP - 1:
                op = "v_subrev_u32" if self.reverse else "v_add_u32"
                p.append(Ins(f"{op} v{V_DMAK_CUR}, s{S_KTILE}, v{V_DMAK_CUR}", "valu", [f"v{V_DMAK_CUR}"], [f"v{V_DMAK_CUR}"]))
            g.append((p, "dma"))
        for jj in range(NP):
            so = "0" if jj == 0 else f"s{S_VJ + jj - 1}"
            p = [Ins(f"s_add_u32 m0, s{S_W1024}, {LDS_VREGION + r2 + 4096 * jj}", "salu", [], ["m0", "scc"], w=0.5),
                 Ins(f"buffer_load_dwordx4 v{V_DMAV_CUR}, {sr(S_VRS, 4)}, {so} offen lds", "dma",
                     ["m0", f"v{V_DMAV_CUR}"], [], w=4.0)]
            if jj == NP - 1:
                op = "v_subrev_u32" if self.reverse else "v_add_u32"
                p.append(Ins(f"{op} v{V_DMAV_CUR}, s{S_VTILE}, v{V_DMAV_CUR}", "valu", [f"v{V_DMAV_CUR}"], [f"v{V_DMAV_CUR}"]))
            g.append((p, "dma"))
        return g

    def addr_update(self):
        """Read addresses of the NEXT iteration: K read slot R1' = R2, V read slot R0' = R1; then rotate."""
        ka = [Ins(f"v_add_u32 v{V_KADDR + i}, s{S_R2}, v{V_KBASE + i}", "valu", [f"v{V_KBASE + i}"], [f"v{V_KADDR + i}"])
              for i in range(KS)]
        va = [Ins(f"v_add_u32 v{V_VADDR}, s{S_R1}, v{V_VBASE}", "valu", [f"v{V_VBASE}"], [f"v{V_VADDR}"])]
        t = S_TMP + 3
        rot = [Ins(f"s_mov_b32 s{t}, s{S_R0}", "salu", [], [f"s{t}"], w=0.5),
               Ins(f"s_mov_b32 s{S_R0}, s{S_R1}", "salu", [], [f"s{S_R0}"], w=0.5),
               Ins(f"s_mov_b32 s{S_R1}, s{S_R2}", "salu", [], [f"s{S_R1}"], w=0.5),
               Ins(f"s_mov_b32 s{S_R2}, s{t}", "salu", [f"s{t}"], [f"s{S_R2}"], w=0.5)]
        return ka, va, rot

    # ------------------------------------------------------------------ one iteration
    def mask_check(self, qb):
        """SALU: does tile (j+1) need masking for this q-block?  -> call the mask routine."""
        t = S_TMP
        u = self.uid()
        lines = [
            f"s_add_u32 s{t}, s{S_J}, 1",
        ] + ([f"s_sub_u32 s{t}, s{S_TOP}, s{t}"] if self.reverse else []) + [
            f"s_lshl_b32 s{S_N0}, s{t}, 6",
            f"s_add_u32 s{t}, s{S_N0}, 63",
            f"s_cmp_gt_i32 s{t}, s{S_HIMIN[qb]}",
            f"s_cbranch_scc1 L_domask_{u}_%=",
            f"s_cmp_lt_i32 s{S_N0}, s{S_LOMAX[qb]}",
            f"s_cbranch_scc0 L_nomask_{u}_%=",
            f"L_domask_{u}_%=:",
            f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_MASKFN[qb], 2)}",
            f"L_nomask_{u}_%=:",
        ]
        return lines

    def gen_iteration(self, a0, a1, a2, cfg, fast=None):
        """a0/a1/a2: tiles j, j+1, j+2 are active for this wave.  fast = (r0, r1, r2): the mask-free steady-state copy
        with the LDS ring slots as immediates (no mask checks, no slot bookkeeping)."""
        self.reset_state()
        # ---- streams
        mf1 = (self.qk_mfmas(1) if a1 else []) + (self.pv_mfmas(1) if a0 else [])
        mf2 = (self.qk_mfmas(0) if a2 else []) + (self.pv_mfmas(0) if a1 else [])
        n1_qk = 2 * KS if a1 else 0
        n2_qk = 2 * KS if a2 else 0
        # LDS reads: (ready = global MFMA index after which the register is free, deadline = global index of the
        # first MFMA that consumes it, Ins)
        lds = []
        nm1 = len(mf1)
        if a2:
            for ks in range(KS):
                for kb in range(2):
                    p = ks * 2 + kb
                    ready = p if a1 else -1          # after QK1 MFMA p of phase 1
                    lds.append([ready, nm1 + p, self.k_read(kb, ks, fast)])
        if a1:
            for ks in range(4):
                for d in range(DB):
                    q = ks * DB + d
                    ready = (n1_qk + q) if a0 else -1
                    for ins in self.v_reads(d, ks, fast):
                        lds.append([ready, nm1 + n2_qk + q, ins])
        lds.sort(key=lambda x: (x[1], x[0]))
        val1 = self.softmax(0) if a1 else []
        val2 = self.softmax(1) if a1 else []
        if fast is None:
            misc = self.misc_stream_paged(None) if self.paged else self.misc_stream()
            ka, va, rot = self.addr_update()
        else:
            misc = self.misc_stream_paged(fast) if self.paged else self.misc_stream_fast(fast)
            ka, va, rot = [], [], []
        if self.paged:
            # the page of NEXT iteration's K tile (j + 5): requested now, read after this iteration's closing lgkmcnt(0).  (An
            # outstanding scalar load only makes the counted LDS waits below stricter: "at most n outstanding" still implies
            # that the LDS read n + 1 places back has returned.)
            # (first the rotation: the entry that arrived during the previous iteration is this iteration's K page, the previous
            #  K page - tile j + 3 - this iteration's V page; only then may the next request overwrite BLKN)
            self.raw(f"s_mov_b32 s{S_BLKV}, s{S_BLKK}")
            self.raw(f"s_mov_b32 s{S_BLKK}, s{S_BLKN}")
            for l in self.paged_request([f"s_add_u32 s{S_SUB}, s{S_J}, 5"], S_BLKN):
                self.raw(l)

        # ---- phase 1
        if a1 and fast is None and "maskchk" not in self.ko:
            for l in self.mask_check(0):
                self.raw(l)
        self._phase(mf1, val1, lds, 0, cfg, phase=1, misc=misc, extra=[])
        # ---- phase 2
        if a1 and fast is None and "maskchk" not in self.ko:
            for l in self.mask_check(1):
                self.raw(l)
        base = len(mf1)
        self._phase(mf2, val2, lds, base, cfg, phase=2, misc=misc, extra=ka + va)
        assert not lds, "unissued LDS reads"
        for grp, _ in misc:
            for ins in grp:
                self.emit(ins)
        misc.clear()
        for ins in rot:
            self.emit(ins)
        self.drain_lds()
        if self.paged:
            self.raw("s_waitcnt lgkmcnt(0)")          # the block-table entry requested at the top of the iteration

    def _emit_valu(self, ins):
        if ins.kind == "call_rescale":
            if "valu" in self.ko:
                return
            qb = ins.aux
            lab = ins.txt.split()[1]
            self.emit(Ins(ins.txt, "salu", ["vcc"], []))
            self.raw(f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_RESCFN[qb], 2)}")
            self.out.append(f"{lab}:")
        else:
            self.emit(ins)

    def _phase(self, mfs, valu, lds, gbase, cfg, phase, misc, extra):
        """Interleave: after each MFMA a budget of filler weight is spent on (1) LDS reads whose register
        is free, (2) DMA / bookkeeping groups at the configured gaps, (3) the VALU stream."""
        valu = list(valu) + list(extra)
        M = len(mfs)
        lead = cfg.get("lds_lead", 3)
        if M == 0:
            for it in [x for x in lds if x[0] < gbase]:
                self.emit(it[2])
                lds.remove(it)
            for ins in valu:
                self._emit_valu(ins)
            return
        dma_gaps = cfg["dma_gaps"].get(phase, [])
        lds_per_gap = cfg.get("lds_per_gap", 1)
        vi = 0
        nv = len(valu)
        for k, m in enumerate(mfs):
            g = gbase + k
            assert all(x[1] > g for x in lds), "an LDS read is scheduled after its consumer"
            self.emit(m)
            # (1) LDS reads whose register is free (ready <= g), earliest deadline first; the count is the
            # rate that still meets every deadline `lead` gaps early, at least lds_per_gap
            rdy = [x for x in lds if x[0] <= g]
            need = 0
            for i, x in enumerate(sorted(lds, key=lambda x: x[1])):
                slack = x[1] - lead - g              # gaps (incl. this one) left to issue reads 0..i
                need = max(need, -(-(i + 1) // max(1, slack)))
            n = min(len(rdy), max(need, lds_per_gap))
            for x in rdy[:n]:
                self.emit(x[2])
                lds.remove(x)
            # (2) DMA / bookkeeping groups
            if k in dma_gaps and misc:
                cnt = dma_gaps.count(k)
                for _ in range(cnt):
                    if misc:
                        grp, _tag = misc.pop(0)
                        for ins in grp:
                            self.emit(ins)
            # (3) VALU: spread what is left evenly over the remaining gaps
            left_gaps = M - k
            take = -(-(nv - vi) // left_gaps)
            cap = cfg.get("valu_cap", 99)
            take = min(take, cap) if k < M - 1 else nv - vi
            for _ in range(take):
                if vi < nv:
                    self._emit_valu(valu[vi])
                    vi += 1
        while vi < nv:
            self._emit_valu(valu[vi])
            vi += 1

    # ------------------------------------------------------------------ routines
    def gen_mask_routine(self, qb):
        o = []
        S = V_S[qb]
        T = V_T
        tlo, tinf = f"v{T}", f"v{T + 1}"
        o.append(f"v_subrev_u32 {tlo}, s{S_N0}, v{V_LOG[qb]}")          # lo_t = (lo - 4g) - n0
        o.append(f"v_mov_b32 {tinf}, 0xff800000")
        o.append("s_nop 0")
        st = S_SUB                                                       # s62, s63: idle after the set-up (the caller's temps are s56..s59)
        tri = self.tri and not getattr(self, "timers_build", False)
        for kb in range(2):
            # wave-uniform state of the 32-key sub-block: every row sees all of it (nothing to do - on a causal diagonal
            # that is one of the two sub-blocks of a tile), no row sees any of it (fill), or mixed (test every element)
            u = self.uid()
            o.append(f"s_add_u32 s{st}, s{S_N0}, {32 * kb}")
            o.append(f"s_add_u32 s{st + 1}, s{st}, 31")
            o.append(f"s_cmp_gt_i32 s{st}, s{S_HIMAX[qb]}")
            o.append(f"s_cbranch_scc1 L_mfill{u}_%=")
            if tri:
                # the ALIGNED diagonal sub-block of a causal-like mask: it starts at the first row's last visible key
                # (st == HIMIN), every row sees one key more than the row above (HIMAX - HIMIN == 31: no clipping at the key
                # tail, no row past the sequence) and no left window cuts into it - then "key position > row" is the same
                # for every pass and tile: ONE v_add of the lane-constant triangle (0 / -inf) per element instead of the
                # sub / compare / select chain (this routine runs on one wave while the other three wait at the barrier)
                o.append(f"s_cmp_eq_u32 s{st}, s{S_HIMIN[qb]}")
                o.append(f"s_cbranch_scc0 L_mnotri{u}_%=")
                o.append(f"s_sub_u32 s{S_TOP}, s{S_HIMAX[qb]}, s{S_HIMIN[qb]}")      # (s61: the ALiBi variant's, idle here)
                o.append(f"s_cmp_eq_u32 s{S_TOP}, 31")
                o.append(f"s_cbranch_scc0 L_mnotri{u}_%=")
                o.append(f"s_cmp_ge_i32 s{st}, s{S_LOMAX[qb]}")
                o.append(f"s_cbranch_scc0 L_mnotri{u}_%=")
                for r in range(16):
                    o.append(f"v_add_f32 v{S + 16 * kb + r}, v{S + 16 * kb + r}, v{V_TRI + r}")
                o.append(f"s_branch L_mnext{u}_%=")
                o.append(f"L_mnotri{u}_%=:")
            o.append(f"s_cmp_gt_i32 s{st + 1}, s{S_HIMIN[qb]}")
            o.append(f"s_cbranch_scc1 L_mpart{u}_%=")
            o.append(f"s_cmp_lt_i32 s{st}, s{S_LOMAX[qb]}")
            o.append(f"s_cbranch_scc0 L_mnext{u}_%=")
            o.append(f"L_mpart{u}_%=:")
            for r in range(16):
                c = 32 * kb + (r & 3) + 8 * (r >> 2)
                t = f"v{T + 2 + (r & 3)}"
                o.append(f"v_sub_u32 {t}, {c}, {tlo}")
                o.append(f"v_cmp_gt_u32 vcc, {t}, v{V_WIDTH[qb]}")
                o.append(f"v_cndmask_b32 v{S + 16 * kb + r}, v{S + 16 * kb + r}, {tinf}, vcc")
            o.append(f"s_branch L_mnext{u}_%=")
            o.append(f"L_mfill{u}_%=:")
            for r in range(16):
                o.append(f"v_mov_b32 v{S + 16 * kb + r}, {tinf}")
            o.append(f"L_mnext{u}_%=:")
        o.append(f"s_setpc_b64 {sr(S_RET, 2)}")
        return o

    def gen_rescale_routine(self, qb):
        o = []
        T = V_T + 8            # temps v60.. : must not collide with the caller's live temps (max tree result is in T+23+qb)
        mxr = f"v{V_T + 23 + qb}"
        mrun, negm, l = f"v{V_MRUN[qb]}", f"v{V_NEGM[qb]}", f"v{V_L[qb]}"
        mxs, mnew, muse, al = f"v{T}", f"v{T + 1}", f"v{T + 2}", f"v{T + 3}"
        o.append("s_nop 7")
        o.append("s_nop 7")
        o.append(f"v_mul_f32 {mxs}, s{S_C}, {mxr}")
        o.append(f"v_max_f32 {mnew}, {mrun}, {mxs}")
        o.append(f"v_max_f32 {muse}, 0xff7fffff, {mnew}")
        o.append(f"v_sub_f32 {al}, {mrun}, {muse}")
        o.append(f"v_exp_f32 {al}, {al}")
        o.append(f"v_mov_b32 {mrun}, {mnew}")
        o.append(f"v_sub_f32 {negm}, 0, {muse}")
        o.append(f"v_mul_f32 {l}, {l}, {al}")
        o.append(f"v_mul_f32 v{V_L2[qb]}, v{V_L2[qb]}, {al}")
        o.append(f"v_add_f32 v{V_THR[qb]}, 0x41000000, {mnew}")               # (m_run + 8) / c
        o.append(f"v_mul_f32 v{V_THR[qb]}, s{S_RC}, v{V_THR[qb]}")
        for i in range(0, 16 * DB, 4):
            for e in range(4):
                o.append(f"v_accvgpr_read_b32 v{T + 4 + e}, a{A_O[qb] + i + e}")
            for e in range(4):
                o.append(f"v_mul_f32 v{T + 4 + e}, v{T + 4 + e}, {al}")
            for e in range(4):
                o.append(f"v_accvgpr_write_b32 a{A_O[qb] + i + e}, v{T + 4 + e}")
        o.append("s_nop 1")
        o.append(f"s_setpc_b64 {sr(S_RET, 2)}")
        return o

    def skew(self, cfg):
        """After a barrier the four waves of the workgroup run the same instruction stream in lock step, so every
        LDS read and every LDS-DMA issue of a wave collides with the other three at the one LDS / texture-address
        unit of the CU.  A wave-dependent delay (wave w waits w steps) right behind the barrier staggers them."""
        n = cfg.get("skew", 0)
        if n <= 0:
            return []
        u = self.uid()
        o = []
        for w in range(3):
            o.append(f"s_cmp_eq_u32 s{S_W1024}, {1024 * w}")
            o.append(f"s_cbranch_scc1 L_skew{u}_%=")
            k = n
            while k > 0:
                o.append(f"s_nop {min(k, 8) - 1}")
                k -= min(k, 8)
        o.append(f"L_skew{u}_%=:")
        return o

    # ------------------------------------------------------------------ whole body
    def gen_dma_tile(self, which, tile_s, slot_s, tmp):
        """prologue LDS-DMA of one K or V tile: tile index in SGPR tile_s, slot offset SGPR slot_s."""
        o = []
        rs, tb, s16, vo, reg = ((S_KRS, S_KTILE, S_K16, V_DMAK, 0) if which == "k" else
                                (S_VRS, S_VTILE, S_V16, V_DMAV, LDS_VREGION))
        if self.paged:
            # synchronous lookup (the prologue is not hot), then the tile's own descriptor
            o += self.paged_request([f"s_mov_b32 s{S_SUB}, s{tile_s}"], S_BLKN)
            o.append("s_waitcnt lgkmcnt(0)")
            o += [i.txt for i in self.paged_desc(which, tile_s, S_BLKN)]
            o.append(f"s_add_u32 s{tmp + 1}, s{slot_s}, s{S_W1024}")
            for jj in range(NP):
                o.append(f"s_add_u32 m0, s{tmp + 1}, {reg + 4096 * jj}")
                o.append("s_nop 0")
                so = "0" if jj == 0 else f"s{(S_KJ if which == 'k' else S_VJ) + jj - 1}"
                o.append(f"buffer_load_dwordx4 v{vo}, {sr(rs, 4)}, {so} offen lds")
            return o
        if self.reverse:
            o.append(f"s_sub_u32 s{tmp}, s{S_TOP}, s{tile_s}")
            o.append(f"s_mul_i32 s{tmp}, s{tmp}, s{tb}")
        else:
            o.append(f"s_mul_i32 s{tmp}, s{tile_s}, s{tb}")
        o.append(f"s_cmp_lt_i32 s{tile_s}, s{S_NMAX}")
        o.append(f"s_cselect_b32 s{tmp}, s{tmp}, s{S_OOB}")
        o.append(f"s_add_u32 s{tmp + 1}, s{slot_s}, s{S_W1024}")
        for jj in range(NP):
            o.append(f"s_add_u32 m0, s{tmp + 1}, {reg + 4096 * jj}")
            o.append("s_nop 0")
            o.append(f"buffer_load_dwordx4 v{vo}, {sr(rs, 4)}, s{tmp} offen lds")
            o.append(f"s_add_u32 s{tmp}, s{tmp}, s{s16}")
        return o

    def gen_body(self, cfg):
        L = []
        A = L.append
        timers = cfg.get("timers", 0)      # measurement build: s_memtime stamps land in the LSE rows r0 .. r0+4 of the wave
        self.timers_build = bool(timers)   # (its stamps live in v246..v251: no triangle block)
        self.tri = self.tri and bool(cfg.get("tri", True))      # (--cfg={"tri":0}: the compare / select chain on every masked tile, A/B)

        def stamp(i):
            if timers:
                A(f"s_memtime {sr(S_TMP + 2, 2)}")
                A("s_waitcnt lgkmcnt(0)")
                A(f"v_mov_b32 v{246 + i}, s{S_TMP + 2}")
        A("s_nop 7")
        stamp(0)
        if timers:
            A(f"v_mov_b32 v250, 0")        # generic iterations
            A(f"v_mov_b32 v251, 0")        # fast-loop iterations
        # ---- routine addresses
        A(f"s_getpc_b64 {sr(S_SUB, 2)}")
        A("L_pc_%=:")
        for (reg, lab) in ((S_MASKFN[0], "L_mask0"), (S_MASKFN[1], "L_mask1"), (S_RESCFN[0], "L_resc0"), (S_RESCFN[1], "L_resc1")):
            A(f"s_add_u32 s{reg}, s{S_SUB}, {lab}_%=-L_pc_%=")
            A(f"s_addc_u32 s{reg + 1}, s{S_SUB + 1}, 0")
        A(f"s_mov_b32 s{S_J}, s{S_JIN}")
        if self.reverse:
            A(f"s_sub_u32 s{S_TOP}, s{S_NMAX}, 1")
        A(f"s_mov_b32 s{S_OOB}, 0x80000000")
        A(f"s_mov_b32 s{S_R0}, 0")
        A(f"s_mov_b32 s{S_R1}, {LDS_STAGE}")
        A(f"s_mov_b32 s{S_R2}, {2 * LDS_STAGE}")
        for jj in range(1, 4):
            A(f"s_mul_i32 s{S_KJ + jj - 1}, s{S_K16}, {jj}")
            A(f"s_mul_i32 s{S_VJ + jj - 1}, s{S_V16}, {jj}")
        A("s_barrier")                                   # previous pass: every wave is done with the LDS ring
        stage_q = HD == 128 and cfg.get("stage_q", True)
        QST = 6 * LDS_STAGE                              # Q staging area behind the ring: 16 KiB per wave
        if stage_q:
            # ---- Q: the wave's 64 rows as whole 256-byte rows by LDS-DMA (16 pieces of 4 rows) into a wave-private staging
            # area in the K image's layout, read back as B fragments below.  (Straight buffer loads in the fragment layout
            # fetch 16-byte shreds of 32 rows per instruction: the prologue of a pass took 6-9 k cycles.)
            sb, so, s16r, s4p = S_N0, S_SUB, S_SUB + 1, S_RET
            A(f"s_lshl_b32 s{sb}, s{S_W1024}, 4")                   # wave * 16384
            A(f"s_add_u32 s{sb}, s{sb}, {QST}")
            A(f"v_readfirstlane_b32 s{so}, v{V_R0}")
            A(f"s_lshl_b32 s{s16r}, s{S_QRB}, 4")                   # 16 rows
            A("s_nop 2")
            A(f"s_mul_i32 s{so}, s{so}, s{S_QRB}")                  # byte offset of the wave's first row
            for pp in range(4):
                A(f"s_mul_i32 s{s4p}, s{S_QRB}, {4 * pp}")
                A(f"v_xor_b32 v{V_T + pp}, {64 * pp}, v{V_QCB}")
                A(f"v_add3_u32 v{V_T + pp}, v{V_T + pp}, v{V_QROW}, s{s4p}")
            for pc in range(16):
                A(f"s_add_u32 m0, s{sb}, {1024 * pc}")
                A("s_nop 0")
                A(f"buffer_load_dwordx4 v{V_T + pc % 4}, {sr(S_QRS, 4)}, s{so} offen lds")
                if pc % 4 == 3:
                    A(f"s_add_u32 s{so}, s{so}, s{s16r}")
        else:
            # ---- Q fragments -> AGPRs
            for qb in range(2):
                for ks in range(KS):
                    A(f"buffer_load_dwordx4 {ar(A_Q[qb] + 4 * ks, 4)}, v{V_QOFF[qb]}, {sr(S_QRS, 4)}, 0 offen offset:{32 * ks}")
        # ---- first tiles: K(n_min) -> R1, V(n_min) -> R1, K(n_min+1) -> R2      (n_min = j_start + 2)
        t = S_TMP
        A(f"s_add_u32 s{t}, s{S_J}, 2")
        L += self.gen_dma_tile("k", t, S_R1, t + 1)
        L += self.gen_dma_tile("v", t, S_R1, t + 1)
        A(f"s_add_u32 s{t}, s{S_J}, 3")
        L += self.gen_dma_tile("k", t, S_R2, t + 1)
        if self.paged:
            # page registers for the first iteration (j = n_min - 2): its V tile is n_min + 1 (BLKK rotates into BLKV), its K
            # tile n_min + 2 (BLKN rotates into BLKK); gen_dma_tile left the page of tile n_min + 1 in BLKN
            A(f"s_mov_b32 s{S_BLKK}, s{S_BLKN}")
            L += self.paged_request([f"s_add_u32 s{S_SUB}, s{S_J}, 4"], S_BLKN)
            A("s_waitcnt lgkmcnt(0)")
        # ---- state
        for i in range(2 * 16 * DB):
            A(f"v_accvgpr_write_b32 a{i}, 0")
        for qb in range(2):
            A(f"v_mov_b32 v{V_MRUN[qb]}, 0xff800000")
            A(f"v_mov_b32 v{V_NEGM[qb]}, 0x7f7fffff")
            A(f"v_mov_b32 v{V_L[qb]}, 0")
            A(f"v_mov_b32 v{V_L2[qb]}, 0")
            A(f"v_mov_b32 v{V_THR[qb]}, 0xff800000")
        for i in range(KS):
            A(f"v_add_u32 v{V_KADDR + i}, s{S_R1}, v{V_KBASE + i}")
        A(f"v_add_u32 v{V_VADDR}, s{S_R0}, v{V_VBASE}")
        if self.alibi:
            import struct
            for r in range(16):
                cr = float((r & 3) + 8 * (r >> 2))
                A(f"v_mul_f32 v{V_C0 + r}, 0x{struct.unpack('<I', struct.pack('<f', cr))[0]:08x}, v{V_BETA}")
        if self.tri and not timers:
            # the triangle of an aligned diagonal sub-block: register r holds key position (r & 3) + 8 (r >> 2) + 4 g of the
            # lane's row l31 - masked (-inf) where the position is past the row
            T = V_T
            A(f"v_mbcnt_lo_u32_b32 v{T}, -1, 0")
            A(f"v_mbcnt_hi_u32_b32 v{T}, -1, v{T}")
            A(f"v_and_b32 v{T + 1}, 31, v{T}")
            A(f"v_lshrrev_b32 v{T + 2}, 5, v{T}")
            A(f"v_lshlrev_b32 v{T + 2}, 2, v{T + 2}")
            A(f"v_sub_u32 v{T + 1}, v{T + 1}, v{T + 2}")             # l31 - 4 g (signed)
            A(f"v_mov_b32 v{T + 3}, 0xff800000")
            for r in range(16):
                A(f"v_cmp_gt_i32 vcc, {(r & 3) + 8 * (r >> 2)}, v{T + 1}")
                A(f"v_cndmask_b32 v{V_TRI + r}, 0, v{T + 3}, vcc")
        A(f"s_waitcnt vmcnt({2 * NP})")                  # Q and K(n_min) have landed
        if stage_q:
            A(f"s_lshl_b32 s{S_N0}, s{S_W1024}, 4")
            A(f"s_add_u32 s{S_N0}, s{S_N0}, {QST}")
            for ks in range(KS):
                A(f"v_add_u32 v{V_T + ks}, s{S_N0}, v{V_KBASE + ks}")
            for qb in range(2):
                for ks in range(KS):
                    A(f"ds_read_b128 {ar(A_Q[qb] + 4 * ks, 4)}, v{V_T + ks} offset:{8192 * qb}")
            A("s_waitcnt lgkmcnt(0)")
        stamp(1)

        # ---- iteration loop + dispatch
        A("L_top_%=:")
        if timers:
            A("v_add_u32 v250, 1, v250")
        if "bar" not in self.ko:
            A("s_barrier")
        L += self.skew(cfg)
        t1, t2 = S_TMP, S_TMP + 1
        # ---- mask-free steady state: three unrolled copies, LDS ring slots as immediates
        slots = [(0, LDS_STAGE, 2 * LDS_STAGE), (LDS_STAGE, 2 * LDS_STAGE, 0), (2 * LDS_STAGE, 0, LDS_STAGE)]
        if cfg.get("fast", True):
            A(f"s_cmp_ge_i32 s{S_J}, s{S_FASTLO}")
            A("s_cbranch_scc0 L_generic_%=")
            A(f"s_cmp_lt_i32 s{S_J}, s{S_FASTEND}")
            A("s_cbranch_scc0 L_generic_%=")
            if not self.paged:
                A(f"s_add_u32 s{t1}, s{S_J}, 4")
                if self.reverse:
                    A(f"s_sub_u32 s{t1}, s{S_TOP}, s{t1}")
                A(f"s_mul_i32 s{t1}, s{t1}, s{S_KTILE}")
                A(f"v_add_u32 v{V_DMAK_CUR}, s{t1}, v{V_DMAK}")
                A(f"s_add_u32 s{t1}, s{S_J}, 3")
                if self.reverse:
                    A(f"s_sub_u32 s{t1}, s{S_TOP}, s{t1}")
                A(f"s_mul_i32 s{t1}, s{t1}, s{S_VTILE}")
                A(f"v_add_u32 v{V_DMAV_CUR}, s{t1}, v{V_DMAV}")
            A(f"s_cmp_eq_u32 s{S_R0}, 0")
            A("s_cbranch_scc1 L_fast0_body_%=")
            A(f"s_cmp_eq_u32 s{S_R0}, {LDS_STAGE}")
            A("s_cbranch_scc1 L_fast1_body_%=")
            A("s_branch L_fast2_body_%=")
        A("L_generic_%=:")
        A(f"s_add_u32 s{t2}, s{S_J}, 2")
        A(f"s_cmp_ge_i32 s{S_J}, s{S_WLO}")
        A("s_cbranch_scc0 L_notfull_%=")
        A(f"s_cmp_lt_i32 s{t2}, s{S_WHI}")
        A("s_cbranch_scc1 L_v111_%=")
        A("L_notfull_%=:")
        # a2 = wlo <= j+2 < whi ; a1 ; a0 -> 3-bit code in s_t1
        A(f"s_add_u32 s{t1}, s{S_J}, 1")
        code = S_TMP + 2
        A(f"s_mov_b32 s{code}, 0")
        for (reg, bit) in ((S_J, 1), (t1, 2), (t2, 4)):
            u = self.uid()
            A(f"s_cmp_ge_i32 s{reg}, s{S_WLO}")
            A(f"s_cbranch_scc0 L_c{u}_%=")
            A(f"s_cmp_lt_i32 s{reg}, s{S_WHI}")
            A(f"s_cbranch_scc0 L_c{u}_%=")
            A(f"s_or_b32 s{code}, s{code}, {bit}")
            A(f"L_c{u}_%=:")
        variants = [(0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 1, 0), (1, 0, 0)]   # (a0, a1, a2); (1,1,1) handled above
        for (a0, a1, a2) in variants:
            A(f"s_cmp_eq_u32 s{code}, {a0 + 2 * a1 + 4 * a2}")
            A(f"s_cbranch_scc1 L_v{a0}{a1}{a2}_%=")
        A("s_branch L_v000_%=")
        report = {}
        for (a0, a1, a2) in [(1, 1, 1)] + variants + [(0, 0, 0)]:
            A(f"L_v{a0}{a1}{a2}_%=:")
            self.out = []
            self.stats = {"nop_states": 0, "lgkm_waits": 0}
            self.gen_iteration(a0, a1, a2, cfg)
            report[(a0, a1, a2)] = (dict(self.stats), len(self.out))
            L += self.out
            A(f"s_waitcnt vmcnt({2 * NP})")
            A(f"s_add_u32 s{S_J}, s{S_J}, 1")
            A(f"s_cmp_lt_i32 s{S_J}, s{S_NMAX}")
            A("s_cbranch_scc1 L_top_%=")
            A("s_branch L_done_%=")
        # ---- fast copies
        if cfg.get("fast", True):
            for c, sl in enumerate(slots):
                A(f"L_fast{c}_%=:")
                if "bar" not in self.ko:
                    A("s_barrier")
                L += self.skew(cfg)
                A(f"L_fast{c}_body_%=:")
                if timers:
                    A("v_add_u32 v251, 1, v251")
                self.out = []
                self.stats = {"nop_states": 0, "lgkm_waits": 0}
                self.gen_iteration(1, 1, 1, cfg, fast=sl)
                report[("fast", c)] = (dict(self.stats), len(self.out))
                L += self.out
                A(f"s_waitcnt vmcnt({2 * NP})")
                A(f"s_add_u32 s{S_J}, s{S_J}, 1")
                A(f"s_cmp_lt_i32 s{S_J}, s{S_FASTEND}")
                if c < 2:
                    A(f"s_cbranch_scc0 L_fastexit{c}_%=")
                else:
                    A("s_cbranch_scc1 L_fast0_%=")
                    A("s_branch L_fastexit2_%=")
            for c, sl in enumerate(slots):
                n0, n1, n2 = sl[1], sl[2], sl[0]           # ring state of the iteration after copy c
                A(f"L_fastexit{c}_%=:")
                A(f"s_mov_b32 s{S_R0}, {n0}")
                A(f"s_mov_b32 s{S_R1}, {n1}")
                A(f"s_mov_b32 s{S_R2}, {n2}")
                for i in range(KS):
                    A(f"v_add_u32 v{V_KADDR + i}, {n1}, v{V_KBASE + i}")
                A(f"v_add_u32 v{V_VADDR}, {n0}, v{V_VBASE}")
                A("s_branch L_top_%=")
        # ---- routines
        for qb in range(2):
            A(f"L_mask{qb}_%=:")
            L += self.gen_mask_routine(qb)
            A(f"L_resc{qb}_%=:")
            L += self.gen_rescale_routine(qb)
        # ---- epilogue: O / l -> 16 bit, LSE.  O goes through a wave-private LDS image ([64 rows][256 B + 16]) so that the
        # stores cover whole 256-byte rows (4 rows per instruction) instead of 8-byte shreds of 32 rows
        EP_QB = 32 * EP_PITCH
        assert (2 * EP_QB) % 1024 == 0
        rsh = {4: 4, 8: 3}[EP_ROWS]                      # lane -> (row, 16-byte chunk) of a 1-KiB store
        A("L_done_%=:")
        A("s_waitcnt vmcnt(0) lgkmcnt(0)")
        stamp(2)
        A("s_barrier")                                   # every wave is done with the K / V ring
        T = V_T
        t = S_TMP
        E = V_P[0]                 # eight lane-derived values in the (dead) P registers: v77..v79 belong to the ALiBi variant
        lane, wbase, rbase, goff = E, E + 1, E + 2, E + 3
        A(f"v_mbcnt_lo_u32_b32 v{lane}, -1, 0")
        A(f"v_mbcnt_hi_u32_b32 v{lane}, -1, v{lane}")
        A(f"v_and_b32 v{E + 4}, 31, v{lane}")                      # q row of the accumulator columns
        A(f"v_lshrrev_b32 v{E + 5}, 5, v{lane}")                   # g
        A(f"v_lshrrev_b32 v{E + 6}, {rsh}, v{lane}")               # row of the lane's 16-byte chunk
        A(f"v_and_b32 v{E + 7}, {(1 << rsh) - 1}, v{lane}")
        A(f"v_lshlrev_b32 v{E + 7}, 4, v{E + 7}")                  # its column byte
        A(f"s_mul_i32 s{t}, s{S_W1024}, {2 * EP_QB // 1024}")      # wave * 17408
        A(f"v_mul_u32_u24 v{wbase}, {EP_PITCH}, v{E + 4}")
        A(f"v_lshl_add_u32 v{wbase}, v{E + 5}, 3, v{wbase}")
        A(f"v_add_u32 v{wbase}, s{t}, v{wbase}")                   # write base: row * pitch + 8 g
        A(f"v_mul_u32_u24 v{rbase}, {EP_PITCH}, v{E + 6}")
        A(f"v_add3_u32 v{rbase}, v{rbase}, v{E + 7}, s{t}")        # read base: row * pitch + column byte
        A(f"v_readfirstlane_b32 s{t + 1}, v{V_ORB}")
        A(f"v_readfirstlane_b32 s{t + 2}, v{V_R0}")
        A("s_nop 4")
        A(f"v_add_u32 v{E + 6}, s{t + 2}, v{E + 6}")               # global row
        A(f"v_mad_u32_u24 v{goff}, v{E + 6}, s{t + 1}, v{E + 7}")  # byte offset of the lane's chunk
        split_ep = (HD == 128) and not self.alibi and not self.paged
        if split_ep:
            A(f"s_cmp_eq_u32 s{S_P32}, 0")
            A("s_cbranch_scc0 L_ep32_%=")
        A(f"s_lshl_b32 s{t + 1}, s{t + 1}, {2 if EP_ROWS == 4 else 3}")   # EP_ROWS rows further
        A("s_nop 7")
        for qb in range(2):
            l, mrun = f"v{V_L[qb]}", f"v{V_MRUN[qb]}"
            ta, lt, inv, lse, zero = f"v{T}", f"v{T + 1}", f"v{T + 2}", f"v{T + 3}", f"v{T + 4}"
            A(f"v_add_f32 {l}, {l}, v{V_L2[qb]}")
            A(f"v_mov_b32 {ta}, {l}")
            A("s_nop 1")
            A(f"v_permlane32_swap_b32 {ta}, {l}")
            A(f"v_add_f32 {lt}, {ta}, {l}")
            A(f"v_rcp_f32 {inv}, {lt}")
            A(f"v_log_f32 {lse}, {lt}")
            A(f"v_mov_b32 {zero}, 0")
            A(f"v_cmp_lt_f32 vcc, 0, {lt}")
            A(f"v_cndmask_b32 {inv}, {zero}, {inv}, vcc")
            A(f"v_add_f32 {lse}, {lse}, {mrun}")
            A(f"v_mul_f32 {lse}, 0x3f317218, {lse}")
            A(f"buffer_store_dword {lse}, v{V_LSEOFF[qb]}, {sr(S_LRS, 4)}, 0 offen")
            for d in range(DB):
                for r4 in range(4):
                    base = A_O[qb] + 16 * d + 4 * r4
                    tt = T + 8 + 4 * (r4 & 1)
                    for e in range(4):
                        A(f"v_accvgpr_read_b32 v{tt + e}, a{base + e}")
                    for e in range(4):
                        A(f"v_mul_f32 v{tt + e}, v{tt + e}, {inv}")
                    pk = V_S[1] + 2 * ((4 * d + r4) % 8)
                    A(f"{self.cvt} v{pk}, v{tt}, v{tt + 1}")
                    A(f"{self.cvt} v{pk + 1}, v{tt + 2}, v{tt + 3}")
                    A(f"ds_write_b64 v{wbase}, {vr(pk, 2)} offset:{qb * EP_QB + 64 * d + 16 * r4}")
        A("s_waitcnt lgkmcnt(0)")
        A(f"s_mov_b32 s{t + 3}, 0")
        NST = 32 // EP_ROWS                              # store instructions per q-block
        for qb in range(2):
            for j in range(NST):
                A(f"ds_read_b128 {vr(V_S[0] + 4 * j, 4)}, v{rbase} offset:{qb * EP_QB + EP_ROWS * EP_PITCH * j}")
            for j in range(NST):
                A(f"s_waitcnt lgkmcnt({NST - 1 - j})")
                A(f"buffer_store_dwordx4 {vr(V_S[0] + 4 * j, 4)}, v{goff}, {sr(S_ORS, 4)}, s{t + 3} offen")
                A(f"s_add_u32 s{t + 3}, s{t + 3}, s{t + 1}")
            A("s_nop 1")
        # (no wait for the stores: their data left the registers at issue, and nothing below reads what they write)
        if split_ep:
            # ---- epilogue of a PART of a key-split block: normalised O in fp32, 16 bytes per lane straight from the accumulators
            # (lane = query row l31 of the 32-row block, registers = d: 32 d + 8 r4 + 4 g + (0..3) are four consecutive floats of the
            # row), the part's LSE like a whole block's.  32-byte runs per row and instruction: ~2 % of a pass, and the rows are
            # read back from L2 by the merge kernel.
            A("s_branch L_epend_%=")
            A("L_ep32_%=:")
            A(f"v_add_u32 v{E + 6}, s{t + 2}, v{E + 4}")                # global row of the lane's accumulator column (q-block 0)
            A(f"v_lshlrev_b32 v{E + 7}, 4, v{E + 5}")                   # 16 g
            A("s_nop 1")
            A(f"v_mad_u32_u24 v{goff}, v{E + 6}, s{t + 1}, v{E + 7}")   # row * row bytes + 16 g
            A(f"s_lshl_b32 s{t + 3}, s{t + 1}, 5")                      # q-block 1: 32 rows further
            A("s_nop 7")
            for qb in range(2):
                l, mrun = f"v{V_L[qb]}", f"v{V_MRUN[qb]}"
                ta, lt, inv, lse, zero = f"v{T}", f"v{T + 1}", f"v{T + 2}", f"v{T + 3}", f"v{T + 4}"
                A(f"v_add_f32 {l}, {l}, v{V_L2[qb]}")
                A(f"v_mov_b32 {ta}, {l}")
                A("s_nop 1")
                A(f"v_permlane32_swap_b32 {ta}, {l}")
                A(f"v_add_f32 {lt}, {ta}, {l}")
                A(f"v_rcp_f32 {inv}, {lt}")
                A(f"v_log_f32 {lse}, {lt}")
                A(f"v_mov_b32 {zero}, 0")
                A(f"v_cmp_lt_f32 vcc, 0, {lt}")
                A(f"v_cndmask_b32 {inv}, {zero}, {inv}, vcc")
                A(f"v_add_f32 {lse}, {lse}, {mrun}")
                A(f"v_mul_f32 {lse}, 0x3f317218, {lse}")
                A(f"buffer_store_dword {lse}, v{V_LSEOFF[qb]}, {sr(S_LRS, 4)}, 0 offen")
                so = "0" if qb == 0 else f"s{t + 3}"
                for d in range(DB):
                    for r4 in range(4):
                        base = A_O[qb] + 16 * d + 4 * r4
                        tt = T + 8 + 4 * (r4 & 1)                      # (v60..v67 as in the 16-bit epilogue: v70 / v71 hold q-block 1's second row sum)
                        for e in range(4):
                            A(f"v_accvgpr_read_b32 v{tt + e}, a{base + e}")
                        for e in range(4):
                            A(f"v_mul_f32 v{tt + e}, v{tt + e}, {inv}")
                        A("s_nop 0")
                        A(f"buffer_store_dwordx4 {vr(tt, 4)}, v{goff}, {sr(S_ORS, 4)}, {so} offen offset:{128 * d + 32 * r4}")
            A("L_epend_%=:")
        if timers:
            A("s_waitcnt vmcnt(0)")
            stamp(3)
            A("s_mov_b64 exec, 1")
            for i in range(6):
                A(f"buffer_store_dword v{246 + i}, v{V_LSEOFF[0]}, {sr(S_LRS, 4)}, 0 offen offset:{4 * i}")
            A("s_mov_b64 exec, -1")
            A("s_waitcnt vmcnt(0)")
        return L, report


DEFAULT_CFG = {
    # MFMA index within the phase after which one DMA / bookkeeping group is emitted (9 groups: offsets, 4 K, 4 V)
    "dma_gaps": {2: [16, 18, 20, 22, 24, 26, 28, 30, 31]},
    "lds_per_gap": 1,
}


def clobbers(alibi=False, paged=False):
    c = ["memory", "vcc", "scc", "m0"]
    if paged:
        c += [f"s{i}" for i in (S_BLKK, S_BLKV, S_BLKN)]
    c += [f"v{i}" for i in range(37, 256) if not (alibi and i in (V_BK[0], V_BK[1], V_BETA))]      # (v16..v36 are inputs)
    c += [f"a{i}" for i in range(256)]
    c += [f"s{i}" for i in range(S_R0, S_LAST + 1)]
    return c


def main():
    cfg = dict(DEFAULT_CFG)
    ko = frozenset()
    prefix = "FA_FWD_ASM"
    for a in sys.argv[1:]:
        if a == "--d=64":                                      # fa_fwd64_asm_gen.h: the D = 64 bodies
            set_dim(64)
            prefix = "FA_FWD64_ASM"
            cfg["dma_gaps"] = {2: [5, 7, 9, 11, 13]}           # offsets + 2 K + 2 V pieces in a 16-MFMA phase
    for a in sys.argv[1:]:
        if a.startswith("--ko="):
            ko = frozenset(x for x in a[5:].split(",") if x)
        elif a.startswith("--cfg="):
            import json
            cfg.update(json.loads(a[6:]))
    if "dma_gaps" in cfg:
        cfg["dma_gaps"] = {int(k): v for k, v in cfg["dma_gaps"].items()}
    print("// GENERATED by gen_fwd_asm.py - do not edit.  See that script for the schedule and the register map.")
    print("#pragma once")
    print(f"#define {prefix}_LDS_BYTES {6 * LDS_STAGE + (65536 if HD == 128 and cfg.get('stage_q', True) else 0)}")
    for (alibi, paged) in ((False, False), (True, False)) + (((False, True),) if HD == 128 else ()):
        tag = "ALIBI_" if alibi else ("PAGED_" if paged else "")
        for dt in ("bf16", "f16"):
            g = Gen(dt, alibi=alibi, paged=paged)
            g.ko = ko
            body, report = g.gen_body(cfg)
            print(f"#define {prefix}_{tag}BODY_{dt.upper()} \\")
            for ln in body:
                print(f'    "{ln}\\n" \\')
            print('    ""')
            for k, (st, n) in report.items():
                print(f"// {dt} {tag}variant a0a1a2={k}: {n} lines, nop states {st['nop_states']}, lgkmcnt waits {st['lgkm_waits']}")
        cl = ", ".join(f'"{c}"' for c in clobbers(alibi, paged))
        print(f"#define {prefix}_{tag}CLOBBERS {cl}")


if __name__ == "__main__":
    main()
