#!/usr/bin/env python3
"""Generator for the hand-scheduled gfx950 dQ kernel (fa_bwd_dq_asm.hip): D = 128, no bias / dropout.

Replaces the hot loop of the reference's kernel/fused_mha_backward.cu:168-242 (dQ accumulation over the key tiles) for
BASELINE config 2's shape family.  The compiler-scheduled fa_bwd_dq_kernel spends 54 % of its wave time waiting on an
instruction (SQ_WAIT_INST_ANY, profiles/r02_rocprofv3_summary.txt); this body is written out by hand on the forward
generator's machinery (gen_fwd_asm.py: instruction records, gfx950 hazard table, LDS return-order tracking, the
MFMA / VALU / LDS interleaver).

Orientation (the forward's): everything transposed, so that a lane owns ONE query column and the row statistics
(LSE log2e, D = rowsum(dO o O)) are one register per lane:
    S^T  = K Q^T        A = K rows   (ds_read_b128 from the stage's K image),  B = Q fragments  (pinned, AGPR)
    dP^T = V dO^T       A = V rows   (ds_read_b128 from the stage's V image),  B = dO fragments (pinned, AGPR)
    P^T  = exp2(S^T c - lse2),  dS^T = P^T o (dP^T - D)     VALU, 3.5 instructions per element, rounded to 16 bit
                        (the "- D" costs no VALU work: the dP chain starts with ONE rank-1 MFMA, A = ones in three contraction
                         slots, B = -D split exactly into three 16-bit terms, that leaves -D[q] in every row of the column)
    dQ^T += K^T dS^T    A = K^T      (ds_read_b64_tr_b16 from the SAME K image), B = packed dS^T in the C layout
Structure (one wave per SIMD, 512 registers, 4 waves x 64 query rows = 256-row workgroup):
  * a wave owns two 32-row q-blocks that run HALF AN ITERATION OUT OF PHASE over 32-key stages:
        phase A of iteration j : MFMA  S1(j+1), dP1(j+1), dQ1(j)      VALU  grad0(j+1)      (24 MFMAs)
        phase B of iteration j : MFMA  S0(j+2), dP0(j+2), dQ0(j+1)    VALU  grad1(j+1)
    so the VALU stream of a phase never depends on the MFMAs issued beside it, and S / dP need no double buffering;
  * dQ^T accumulators a[0:127], Q fragments a[128:191], dO fragments a[192:255]; the stage's K-row, V-row and K^T
    fragments (3 x 32 VGPRs) are reloaded right behind their use by q-block 1 (phase A) and are consumed by q-block 0
    in phase B and by q-block 1 in the next phase A: every LDS fragment is read ONCE per two q-blocks;
  * a stage = 32 keys: K tile + V tile (8 KiB each, the `swzt` image of fa_common.h that serves row and transposed reads)
    arrive by LDS-DMA in a ring of four 16 KiB slots, issued two iterations ahead behind a counted vmcnt; the loop is
    unrolled by four so that every LDS address is an immediate;
  * q-block 0's S / dP of the first stage are computed in front of the loop and the last iteration ends after phase A;
    what is left of fill and drain works on VIRTUAL stages (zero-filled through an out-of-range DMA source, fully
    masked): one loop body; masks (causal / window / key tail) set S = -inf in a called routine on edge stages only;
  * prologue: Q -> AGPRs, dO and O -> registers, D = rowsum(dO o O) in fp32 (the fused preprocess: this kernel runs
    first and leaves softmax_d and the statistics planes of the asm dK/dV kernel behind), dO -> AGPRs;
  * epilogue: dQ * softmax_scale -> 16 bit through a per-wave LDS image so that each store covers four whole rows.
Causal-like ALiBi variant (DQ(alibi=True)): bias(q, key) = slope (key - off - q) needs no operand here - a lane owns ONE query and
a register ONE key row of the sub-tile, so in log2 units
        S c + slope2 (n0 + kr + 4g - off - q) - lse2  =  S c + slope2 kr - L',    L' = [lse2 + slope2 (off + q - 4g)] - slope2 n0
with kr = (r & 3) + 8 (r >> 2) a literal per register: L' is one v_fma per q-block and stage, the kr term one v_fmamk per element
(15 per q-block and stage, in the MFMAs' shadow: the matrix pipe bounds this kernel); four more registers, no LDS, no MFMA.

Run:  python gen_bwd_dq_asm.py > fa_bwd_dq_asm_gen.h
"""
import sys
from gen_fwd_asm import Ins, Gen, rl, vr, ar, sr

# ------------------------------------------------------------------ LDS map
STG = 16384                     # one stage: K tile 8 KiB + V tile 8 KiB
NRING = 4
EP_PITCH = 272                  # epilogue: [64 rows][256 B + 16] per wave
EP_QB = 32 * EP_PITCH
QST = NRING * STG               # prologue: per wave [Q | dO | O] x [32 rows][256 B] of ONE q-block (24 KiB), behind the ring
QST_WAVE = 3 * 8192
LDS_TOTAL = max(QST + 4 * QST_WAVE, 4 * 2 * EP_QB)

# ------------------------------------------------------------------ SGPRs (inputs s16..s59, owned s60..)
S_QRS, S_DORS, S_ORS, S_KRS, S_VRS, S_DQRS, S_LRS, S_SDRS, S_STRS = 16, 20, 24, 28, 32, 36, 40, 44, 48
S_C, S_SCALE = 52, 53           # softmax_scale * log2e, softmax_scale
S_JIN, S_NMAX, S_NMIN = 54, 55, 56   # first iteration (n_min - 1), stage range [n_min, n_max)
S_KSTG, S_VSTG = 57, 58         # bytes of one 32-key stage of K / V
S_W1024 = 59
S_MLO = (60, 62)                # in: a stage starting at key n0 is free of masks for the q-block iff (n0 - MLO) <=u MRANGE
S_MRANGE = (61, 63)             #     (MLO = max over the rows of the first visible key, MRANGE = min last visible key - 31 - MLO;
                                #      MLO = 0x40000000, MRANGE = 0: every stage is masked)
S_PLANE = 64                    # in: bytes of one statistics plane (0 when there is no statistics workspace)
S_WHI = 65                      # in: this wave has no visible key in stages >= S_WHI ...
S_WLO = 66                      # ... nor in stages < S_WLO: iterations outside [WLO - 2, WHI) only move data
# owned
S_J = 68
S_T = 69                        # s69..s73 temps
S_KSO, S_VSO = 74, 75           # DMA stream: source offsets of stage j + 4
S_OOB = 76
S_N0 = 77
S_SUB, S_RET = 78, 80
S_MASKFN = (82, 84)
S_LAST = 85

# ------------------------------------------------------------------ VGPRs (v0..v15 are left to the compiler)
V_CBS = 16                      # in: v16..v19 = swizzled column byte of the lane's 16-byte DMA chunk for rows 4p + (lane >> 4), p = 0..3
V_QRB, V_DORB = 20, 21          # in (uniform): bytes per Q / dO row
V_LSEOFF = (22, 23)             # in: row * 4 (LSE load, softmax_d / statistics stores)
V_ROW = 24                      # in: 8 row-read addresses (swzt image, tile-relative)
V_TR = 32                       # in: 8 transposed-read addresses [h][d]
V_DMAK, V_DMAV = 40, 41         # in: LDS-DMA source voffsets (first piece; the second lies 16 rows further)
V_LOG = (42, 44)                # in: lo - 4g of the lane's row (0x3fffffff: no row)
V_WID = (43, 45)                # in: hi - lo
V_DQRB, V_R0 = 46, 47           # in (uniform): bytes per dQ row, first row of the wave's 64
V_DMAK2, V_DMAV2 = 48, 49       # in: the second piece of a stage (16 rows further)
V_ORB = 50                      # in (uniform): bytes per O row
# owned (v51 free)
V_LSE2 = (52, 54)               # lse * log2e (+inf for rows without keys)
V_D = (53, 55)                  # rowsum(dO o O)
V_T = 56                        # temps v56..v63
V_ND = (64, 68)                 # B operand of the rank-1 MFMA that starts the dP chain at -D: -D split into three 16-bit
                                # terms in the contraction slots 0..2 (lanes 0..31; zeros elsewhere), 4 regs per q-block
V_ONE = 252                     # its A operand: ones in the contraction slots 0..2 (4 regs)
V_S = (72, 88)                  # S^T accumulators (16 each)
V_DP = (104, 120)               # dP^T accumulators
V_DS = (136, 144)               # packed dS^T (8 each)
V_KR = 152                      # K-row fragments [ks] x 4
V_VR = 184                      # V-row fragments [ks] x 4
V_KT = 216                      # K^T fragments [t][d] x 4           (.. v247)
V_TM = 248                      # measurement build: time stamps (v248..v251)
# causal-like ALiBi variant (no measurement build of it)
V_SL = 51                       # in (uniform): slope * log2(e)
S_OFF = 67                      # in: seqlen_k - seqlen_q
V_LP = (248, 249)               # owned: L' of the stage whose gradients are computed (V_LSE2 holds the bracket above)
V_N0F = 250                     # owned: float(n0)
A_DQ = (0, 64)
A_Q = (128, 160)
A_DO = (192, 224)


class DQ(Gen):
    def __init__(self, dtype, alibi=False):
        Gen.__init__(self, dtype)
        self.alibi_dq = alibi

    def reset_dq(self, mfma_age=8):
        self.now = 0
        self.last = {}
        self.lds_q = []
        self.srcc_rd = {}
        for qb in (0, 1):
            for r in rl("v", V_S[qb], 16) + rl("v", V_DP[qb], 16) + rl("a", A_DQ[qb], 64):
                self.last[r] = (-mfma_age, "mfma", None)

    # ---- MFMA streams of one q-block ----
    def dq_mfmas(self, qb):
        """dQ^T[d] += K^T[t][d] dS^T[t]"""
        out = []
        for t in range(2):
            for d in range(4):
                out.append((("kt", t * 4 + d), self.mfma("a", A_DQ[qb] + 16 * d, "v", V_KT + 4 * (t * 4 + d), "v", V_DS[qb] + 4 * t, False)))
        return out

    def sdp_mfmas(self, qb):
        """S^T (+)= K[ks] Q[ks]^T and dP^T (+)= V[ks] dO[ks]^T, interleaved (two accumulator chains)"""
        out = [(("nd", 0), self.mfma("v", V_DP[qb], "v", V_ONE, "v", V_ND[qb], True))]       # dP := -D (rank-1)
        for ks in range(8):
            out.append((("kr", ks), self.mfma("v", V_S[qb], "v", V_KR + 4 * ks, "a", A_Q[qb] + 4 * ks, ks == 0)))
            out.append((("vr", ks), self.mfma("v", V_DP[qb], "v", V_VR + 4 * ks, "a", A_DO[qb] + 4 * ks, False)))
        return out

    # ---- LDS reads (slot offsets are immediates) ----
    def frag_reads(self, which, idx, slot):
        if which == "nd":
            return []
        if which == "kr":
            b = V_KR + 4 * idx
            return [Ins(f"ds_read_b128 {vr(b, 4)}, v{V_ROW + idx} offset:{slot * STG}", "lds", [f"v{V_ROW + idx}"], rl("v", b, 4))]
        if which == "vr":
            b = V_VR + 4 * idx
            return [Ins(f"ds_read_b128 {vr(b, 4)}, v{V_ROW + idx} offset:{slot * STG + 8192}", "lds", [f"v{V_ROW + idx}"], rl("v", b, 4))]
        t, d = idx // 4, idx % 4
        b = V_KT + 4 * idx
        return [Ins(f"ds_read_b64_tr_b16 {vr(b + 2 * h, 2)}, v{V_TR + 4 * h + d} offset:{slot * STG + t * 4096}", "lds",
                    [f"v{V_TR + 4 * h + d}"], rl("v", b + 2 * h, 2)) for h in range(2)]

    # ---- VALU stream: P = exp2(S c - lse2), dS = P (dP - D), pack ----
    def grad(self, qb):
        S, DP, DS = V_S[qb], V_DP[qb], V_DS[qb]
        al = self.alibi_dq
        l2 = f"v{V_LP[qb]}" if al else f"v{V_LSE2[qb]}"
        out = []
        if al:      # L' of this stage (first key S_N0)
            out.append(Ins(f"v_cvt_f32_i32 v{V_N0F}, s{S_N0}", "valu", [f"s{S_N0}"], [f"v{V_N0F}"]))
            out.append(Ins(f"v_fma_f32 {l2}, v{V_N0F}, -v{V_SL}, v{V_LSE2[qb]}", "valu", [f"v{V_N0F}", f"v{V_SL}", f"v{V_LSE2[qb]}"], [l2]))
        d = 1 if al else 0          # the ALiBi variant has one more step in front of the exponential
        for r in range(16 + 3 + d):
            if r < 16:
                out.append(Ins(f"v_fma_f32 v{S + r}, v{S + r}, s{S_C}, -{l2}", "valu", [f"v{S + r}", l2], [f"v{S + r}"]))
            if al and 0 <= r - 1 < 16:
                q = r - 1
                kr = (q & 3) + 8 * (q >> 2)
                if kr:
                    import struct
                    lit = struct.unpack("<I", struct.pack("<f", float(kr)))[0]
                    out.append(Ins(f"v_fmamk_f32 v{S + q}, v{V_SL}, 0x{lit:08x}, v{S + q}", "valu", [f"v{S + q}", f"v{V_SL}"], [f"v{S + q}"]))
            if 0 <= r - 1 - d < 16:
                q = r - 1 - d
                out.append(Ins(f"v_exp_f32 v{S + q}, v{S + q}", "trans", [f"v{S + q}"], [f"v{S + q}"], w=1.6))
            if 0 <= r - 3 - d < 16:
                q = r - 3 - d
                out.append(Ins(f"v_mul_f32 v{DP + q}, v{S + q}, v{DP + q}", "valu", [f"v{S + q}", f"v{DP + q}"], [f"v{DP + q}"]))
                if q % 2 == 1:
                    e = q // 2
                    out.append(Ins(f"{self.cvt} v{DS + e}, v{DP + q - 1}, v{DP + q}", "valu", [f"v{DP + q - 1}", f"v{DP + q}"], [f"v{DS + e}"]))
        return out

    # ---- DMA of stage j + 4 into ring slot `slot`: 2 K pieces + 2 V pieces per wave ----
    def dma_groups(self, slot):
        g = []
        t = S_T
        for (rs, so, vos, toff) in ((S_KRS, t + 3, (V_DMAK, V_DMAK2), 0), (S_VRS, t + 4, (V_DMAV, V_DMAV2), 8192)):
            for jj in range(2):
                vo = vos[jj]
                g.append(([Ins(f"s_add_u32 m0, s{S_W1024}, {slot * STG + toff + 4096 * jj}", "salu", [], ["m0", "scc"], w=0.5),
                           Ins(f"buffer_load_dwordx4 v{vo}, {sr(rs, 4)}, s{so} offen lds", "dma", ["m0", f"v{vo}", f"s{so}"], [], w=4.0)], "dma"))
        return g

    def mask_check(self, qb):
        """SALU: does stage j + 1 (first key S_N0) need masking for this q-block?  -> call the mask routine."""
        t = S_T
        u = self.uid()
        return [f"s_sub_u32 s{t}, s{S_N0}, s{S_MLO[qb]}",
                f"s_cmp_le_u32 s{t}, s{S_MRANGE[qb]}",
                f"s_cbranch_scc1 L_nomask_{u}_%=",
                f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_MASKFN[qb], 2)}",
                f"L_nomask_{u}_%=:"]

    def gen_mask_routine(self, qb):
        o = []
        S, T = V_S[qb], V_T
        tlo, tinf = f"v{T}", f"v{T + 1}"
        o.append("s_nop 7")                                            # (the S accumulators may have been written 8 states ago)
        o.append(f"v_subrev_u32 {tlo}, s{S_N0}, v{V_LOG[qb]}")         # lo_t = (lo - 4g) - n0
        o.append(f"v_mov_b32 {tinf}, 0xff800000")
        o.append("s_nop 0")
        for r in range(16):
            c = (r & 3) + 8 * (r >> 2)
            t = f"v{T + 2 + (r & 3)}"
            o.append(f"v_sub_u32 {t}, {c}, {tlo}")
            o.append(f"v_cmp_gt_u32 vcc, {t}, v{V_WID[qb]}")
            o.append(f"v_cndmask_b32 v{S + r}, v{S + r}, {tinf}, vcc")
        o.append(f"s_setpc_b64 {sr(S_RET, 2)}")
        return o

    # ---- one iteration (copy c of 4: iteration index i = j - j_start, i % 4 == c) ----
    def gen_iteration(self, c, cfg):
        self.reset_dq()
        slot_rows, slot_tr, slot_dma = (c + 2) % NRING, (c + 1) % NRING, c % NRING
        A = self.raw
        t = S_T
        # SALU head: source offsets of the stage the DMA stream points at (j + 4): zeros past the last real stage
        A(f"s_add_u32 s{t}, s{S_J}, 4")
        A(f"s_cmp_lt_i32 s{t}, s{S_NMAX}")
        A(f"s_cselect_b32 s{t + 3}, s{S_KSO}, s{S_OOB}")
        A(f"s_cselect_b32 s{t + 4}, s{S_VSO}, s{S_OOB}")
        # S / dP first: the VALU stream of the NEXT phase starts on them; dQ (operands ready since the previous phase) last
        mfA = self.sdp_mfmas(1) + self.dq_mfmas(1)
        mfB = self.sdp_mfmas(0) + self.dq_mfmas(0)
        nA = len(mfA)
        lds = []
        for k, (tag, mf) in enumerate(mfA):
            slot = slot_tr if tag[0] == "kt" else slot_rows
            for ins in self.frag_reads(tag[0], tag[1], slot):
                lds.append([k, nA + k, ins])
        lds.sort(key=lambda x: (x[1], x[0]))
        misc = self.dma_groups(slot_dma)
        # ---- phase A: q-block 1 on the matrix pipe, q-block 0's gradient arithmetic (stage j + 1) on the VALU
        for l in self.mask_check(0):
            A(l)
        self._phase([m for _, m in mfA], self.grad(0), lds, 0, cfg, phase=1, misc=misc, extra=[])
        # ---- the last iteration (j = n_max - 1) ends here: its phase B would touch virtual stages only
        A(f"s_add_u32 s{t}, s{S_J}, 1")
        A(f"s_cmp_ge_i32 s{t}, s{S_NMAX}")
        A("s_cbranch_scc1 L_done_%=")
        # ---- phase B
        for l in self.mask_check(1):
            A(l)
        self._phase([m for _, m in mfB], self.grad(1), lds, nA, cfg, phase=2, misc=misc, extra=[])
        assert not lds, "unissued LDS reads"
        for grp, _ in misc:
            for ins in grp:
                self.emit(ins)
        misc.clear()
        self.drain_lds()

    def gen_data_only(self, c):
        """an iteration of a wave that has no visible key in it: its share of the stage DMA, nothing else"""
        self.reset_dq()
        A = self.raw
        t = S_T
        A(f"s_add_u32 s{t}, s{S_J}, 4")
        A(f"s_cmp_lt_i32 s{t}, s{S_NMAX}")
        A(f"s_cselect_b32 s{t + 3}, s{S_KSO}, s{S_OOB}")
        A(f"s_cselect_b32 s{t + 4}, s{S_VSO}, s{S_OOB}")
        A("s_nop 3")
        for grp, _ in self.dma_groups(c % NRING):
            for ins in grp:
                self.emit(ins)

    # ---- whole body ----
    def gen_body(self, cfg):
        L = []
        A = L.append
        t = S_T
        timers = cfg.get("timers", 0)

        def stamp(i):
            if timers:
                A(f"s_memtime {sr(t + 1, 2)}")
                A("s_waitcnt lgkmcnt(0)")
                A(f"v_mov_b32 v{V_TM + i}, s{t + 1}")
        A("s_nop 7")
        stamp(0)
        A(f"s_getpc_b64 {sr(S_SUB, 2)}")
        A("L_pc_%=:")
        for (reg, lab) in ((S_MASKFN[0], "L_mask0"), (S_MASKFN[1], "L_mask1")):
            A(f"s_add_u32 s{reg}, s{S_SUB}, {lab}_%=-L_pc_%=")
            A(f"s_addc_u32 s{reg + 1}, s{S_SUB + 1}, 0")
        A(f"s_mov_b32 s{S_OOB}, 0x80000000")
        A(f"s_mov_b32 s{S_J}, s{S_JIN}")
        A(f"s_add_u32 s{S_N0}, s{S_J}, 1")
        A(f"s_lshl_b32 s{S_N0}, s{S_N0}, 5")                      # first key of stage j + 1
        A("s_barrier")                                            # previous pass is done with LDS
        DOT = (V_KR, V_KT)          # dO / O of q-block 0: K-row / V-row fragment registers, of q-block 1: K^T fragments and
        OT = (V_VR, V_S[0])         # the S accumulators - all dead until the loop
        # ---- stages j+1, j+2, j+3 -> ring slots 1, 2, 3 (stage j + 1 = n_min - 1 is virtual: zeros)
        A(f"s_add_u32 s{t}, s{S_J}, 1")
        A(f"s_mul_i32 s{S_KSO}, s{t}, s{S_KSTG}")
        A(f"s_mul_i32 s{S_VSO}, s{t}, s{S_VSTG}")
        for slot in (1, 2, 3):
            A(f"s_add_u32 s{t}, s{S_J}, {slot}")
            A(f"s_cmp_ge_i32 s{t}, s{S_NMIN}")
            A(f"s_cselect_b32 s{t + 3}, s{S_KSO}, s{S_OOB}")
            A(f"s_cselect_b32 s{t + 4}, s{S_VSO}, s{S_OOB}")
            A(f"s_cmp_lt_i32 s{t}, s{S_NMAX}")
            A(f"s_cselect_b32 s{t + 3}, s{t + 3}, s{S_OOB}")
            A(f"s_cselect_b32 s{t + 4}, s{t + 4}, s{S_OOB}")
            for (rs, so, vos, toff) in ((S_KRS, t + 3, (V_DMAK, V_DMAK2), 0), (S_VRS, t + 4, (V_DMAV, V_DMAV2), 8192)):
                for jj in range(2):
                    A(f"s_add_u32 m0, s{S_W1024}, {slot * STG + toff + 4096 * jj}")
                    A("s_nop 0")
                    A(f"buffer_load_dwordx4 v{vos[jj]}, {sr(rs, 4)}, s{so} offen lds")
            A(f"s_add_u32 s{S_KSO}, s{S_KSO}, s{S_KSTG}")
            A(f"s_add_u32 s{S_VSO}, s{S_VSO}, s{S_VSTG}")
        # (S_KSO / S_VSO now point at stage j + 4: the first iteration's DMA)
        # ---- Q, dO, O: whole 256-byte rows by LDS-DMA into a wave-private staging area (each wave its own 32 rows of one
        # q-block: 3 x 8 pieces of 4 rows), read back as MFMA B fragments (Q -> AGPRs, dO / O -> registers).  Straight
        # buffer loads in the fragment layout fetch 16-byte shreds of 32 rows per instruction: 16-20 k cycles per pass.
        VO = V_DP[0]                # 12 lane offsets [tensor][p]: (4p + (lane >> 4)) * row_bytes + swizzled column byte
        AD = V_DP[1]                # 8 fragment read addresses inside the staging area
        S_P = (t + 2, t + 3, t + 4)  # s: bytes per Q / dO / O row (the stage DMA above is done with these temps)
        S_ROW0 = S_RET              # s: first row of the wave (the call registers are idle until the loop)
        SB = S_RET + 1              # s: staging base of the wave
        A(f"v_mbcnt_lo_u32_b32 v{V_T}, -1, 0")
        A(f"v_mbcnt_hi_u32_b32 v{V_T}, -1, v{V_T}")
        A(f"v_lshrrev_b32 v{V_T}, 4, v{V_T}")                     # lane >> 4
        A(f"v_readfirstlane_b32 s{S_P[0]}, v{V_QRB}")
        A(f"v_readfirstlane_b32 s{S_P[1]}, v{V_DORB}")
        A(f"v_readfirstlane_b32 s{S_P[2]}, v{V_ORB}")
        A(f"v_readfirstlane_b32 s{S_ROW0}, v{V_R0}")
        A("s_nop 4")
        for ti in range(3):
            for pp in range(4):
                A(f"v_add_u32 v{V_T + 1}, {4 * pp}, v{V_T}")
                A(f"v_mad_u32_u24 v{VO + 4 * ti + pp}, v{V_T + 1}, s{S_P[ti]}, v{V_CBS + pp}")
        A(f"s_mul_i32 s{SB}, s{S_W1024}, {QST_WAVE // 1024}")
        A(f"s_add_u32 s{SB}, s{SB}, {QST}")
        for ks in range(8):
            A(f"v_add_u32 v{AD + ks}, s{SB}, v{V_ROW + ks}")

        def stage_rows(qb):
            """this wave's 32 rows of q-block qb: 24 pieces"""
            for ti, rs in enumerate((S_QRS, S_DORS, S_ORS)):
                A(f"s_add_u32 s{t}, s{S_ROW0}, {32 * qb}")
                A(f"s_mul_i32 s{t}, s{t}, s{S_P[ti]}")                    # byte offset of the q-block's first row
                A(f"s_lshl_b32 s{t + 1}, s{S_P[ti]}, 4")
                A(f"s_add_u32 s{t + 1}, s{t + 1}, s{t}")                  # ... and of its 16th
                for pc in range(8):
                    A(f"s_add_u32 m0, s{SB}, {ti * 8192 + 1024 * pc}")
                    A("s_nop 0")
                    A(f"buffer_load_dwordx4 v{VO + 4 * ti + pc % 4}, {sr(rs, 4)}, s{t + pc // 4} offen lds")

        def read_rows(qb):
            for ks in range(8):
                A(f"ds_read_b128 {ar(A_Q[qb] + 4 * ks, 4)}, v{AD + ks}")
                A(f"ds_read_b128 {vr(DOT[qb] + 4 * ks, 4)}, v{AD + ks} offset:8192")
                A(f"ds_read_b128 {vr(OT[qb] + 4 * ks, 4)}, v{AD + ks} offset:16384")
        stage_rows(0)
        for qb in range(2):
            A(f"buffer_load_dword v{V_LSE2[qb]}, v{V_LSEOFF[qb]}, {sr(S_LRS, 4)}, 0 offen")
        # ---- (while the loads are in flight) accumulators 0
        for i in range(128):
            A(f"v_accvgpr_write_b32 a{i}, 0")
        A("s_waitcnt vmcnt(0)")
        read_rows(0)
        A("s_waitcnt lgkmcnt(0)")
        stage_rows(1)                                             # (q-block 0's D is computed while these are in flight)
        # ---- D = rowsum(dO o O), lse2, statistics; dO -> AGPRs
        T = V_T
        for qb in range(2):
            if qb == 1:
                A("s_waitcnt vmcnt(0)")                            # q-block 1's rows have landed (q-block 0's statistics are stored below)
                read_rows(1)
                A("s_waitcnt lgkmcnt(0)")
            acc, acc2 = f"v{V_D[qb]}", f"v{T + 4}"
            A(f"v_mov_b32 {acc}, 0")
            A(f"v_mov_b32 {acc2}, 0")
            for i in range(32):
                o, do = f"v{OT[qb] + i}", f"v{DOT[qb] + i}"
                if self.dtype == "bf16":
                    A(f"v_lshlrev_b32 v{T}, 16, {o}")
                    A(f"v_lshlrev_b32 v{T + 1}, 16, {do}")
                    A(f"v_and_b32 v{T + 2}, 0xffff0000, {o}")
                    A(f"v_and_b32 v{T + 3}, 0xffff0000, {do}")
                    A(f"v_fma_f32 {acc}, v{T}, v{T + 1}, {acc}")
                    A(f"v_fma_f32 {acc2}, v{T + 2}, v{T + 3}, {acc2}")
                else:
                    A(f"v_fma_mix_f32 {acc}, {o}, {do}, {acc} op_sel_hi:[1,1,0]")
                    A(f"v_fma_mix_f32 {acc2}, {o}, {do}, {acc2} op_sel:[1,1,0] op_sel_hi:[1,1,0]")
            A(f"v_add_f32 {acc}, {acc}, {acc2}")
            A(f"v_mov_b32 v{T}, {acc}")
            A("s_nop 1")
            A(f"v_permlane32_swap_b32 v{T}, {acc}")
            A(f"v_add_f32 {acc}, v{T}, {acc}")
            l2 = f"v{V_LSE2[qb]}"
            A(f"v_mov_b32 v{T + 1}, 0x7f800000")
            A(f"v_cmp_eq_f32 vcc, 0xff800000, {l2}")
            A(f"v_mul_f32 {l2}, 0x3fb8aa3b, {l2}")
            A(f"v_cndmask_b32 {l2}, {l2}, v{T + 1}, vcc")             # rows without keys: P = exp2(S c - inf) = 0
            A(f"v_sub_f32 v{T + 2}, 0, {acc}")
            if qb == 1:       # q-block 0's stores were held back: a store completing out of order must not satisfy the DMA wait above
                A(f"buffer_store_dword v{V_D[0]}, v{V_LSEOFF[0]}, {sr(S_SDRS, 4)}, 0 offen")
                A(f"buffer_store_dword v{V_LSE2[0]}, v{V_LSEOFF[0]}, {sr(S_STRS, 4)}, 0 offen")
                A(f"buffer_store_dword v{T + 7}, v{V_LSEOFF[0]}, {sr(S_STRS, 4)}, s{S_PLANE} offen")
                A(f"buffer_store_dword {acc}, v{V_LSEOFF[qb]}, {sr(S_SDRS, 4)}, 0 offen")          # softmax_d
                A(f"buffer_store_dword {l2}, v{V_LSEOFF[qb]}, {sr(S_STRS, 4)}, 0 offen")           # statistics plane 0
                A(f"buffer_store_dword v{T + 2}, v{V_LSEOFF[qb]}, {sr(S_STRS, 4)}, s{S_PLANE} offen")   # plane 1: -D
            else:
                A(f"v_mov_b32 v{T + 7}, v{T + 2}")                 # -D of q-block 0, stored with q-block 1's
            # -D = hi + mid + lo exactly, three 16-bit terms (truncating splits of the fp32 remainder for bf16, rounding ones
            # for fp16: every remainder is exact in fp32); lanes 32..63 hold the contraction slots 8..15: zeros
            nd, h, m, lo_, r = f"v{T + 2}", f"v{T + 3}", f"v{T + 4}", f"v{T + 5}", f"v{T + 6}"
            b0, b1 = f"v{V_ND[qb]}", f"v{V_ND[qb] + 1}"
            if self.dtype == "bf16":
                A(f"v_and_b32 {h}, 0xffff0000, {nd}")
                A(f"v_sub_f32 {r}, {nd}, {h}")
                A(f"v_and_b32 {m}, 0xffff0000, {r}")
                A(f"v_sub_f32 {r}, {r}, {m}")
                A(f"v_lshrrev_b32 {h}, 16, {h}")
                A(f"v_or_b32 {b0}, {h}, {m}")
                A(f"v_lshrrev_b32 {b1}, 16, {r}")
            else:
                # fp16: the first term travels scaled by 2^-12 (its ones-slot holds 4096): -D may exceed the fp16 range -
                # loss-scaled gradients do - as long as |D| < 65504 * 4096 (where dS itself no longer fits the format)
                A(f"v_mul_f32 {lo_}, 0x39800000, {nd}")                      # * 2^-12
                A(f"v_cmp_lt_f32 vcc, |{nd}|, 0.5")                          # small D: the unscaled terms carry all of it (a scaled
                A(f"v_cndmask_b32 {lo_}, {lo_}, 0, vcc")                     # first term would be an fp16 subnormal)
                A(f"v_cvt_f16_f32 {h}, {lo_}")
                A(f"v_cvt_f32_f16 {lo_}, {h}")
                A(f"v_mul_f32 {lo_}, 0x45800000, {lo_}")                     # * 4096: exact
                A(f"v_sub_f32 {r}, {nd}, {lo_}")
                A(f"v_cvt_f16_f32 {m}, {r}")
                A(f"v_cvt_f32_f16 {lo_}, {m}")
                A(f"v_sub_f32 {r}, {r}, {lo_}")
                A(f"v_cvt_f16_f32 {lo_}, {r}")
                A(f"v_pack_b32_f16 {b0}, {h}, {m}")
                A(f"v_and_b32 {b1}, 0xffff, {lo_}")
            A(f"v_mbcnt_lo_u32_b32 {h}, -1, 0")
            A(f"v_mbcnt_hi_u32_b32 {h}, -1, {h}")
            A(f"v_cmp_gt_u32 vcc, 32, {h}")
            A(f"v_cndmask_b32 {b0}, 0, {b0}, vcc")
            A(f"v_cndmask_b32 {b1}, 0, {b1}, vcc")
            A(f"v_mov_b32 v{V_ND[qb] + 2}, 0")
            A(f"v_mov_b32 v{V_ND[qb] + 3}, 0")
            if qb == 0:
                one = 0x3f80 if self.dtype == "bf16" else 0x3c00
                first = one if self.dtype == "bf16" else 0x6c00                 # fp16: 4096.0 (see the split above)
                A(f"v_mov_b32 {m}, 0x{first | (one << 16):08x}")
                A(f"v_mov_b32 {lo_}, 0x{one:08x}")
                A(f"v_cndmask_b32 v{V_ONE}, 0, {m}, vcc")
                A(f"v_cndmask_b32 v{V_ONE + 1}, 0, {lo_}, vcc")
                A(f"v_mov_b32 v{V_ONE + 2}, 0")
                A(f"v_mov_b32 v{V_ONE + 3}, 0")
            for i in range(32):
                A(f"v_accvgpr_write_b32 a{A_DO[qb] + i}, v{DOT[qb] + i}")
        if self.alibi_dq:
            # V_LSE2 := lse2 + slope2 (off + q - 4g) (the statistics stores above have read lse2; rows past the sequence carry the
            # out-of-range marker as their offset: a huge bracket, P = 0 - they are masked anyway)
            assert not timers
            A(f"v_mbcnt_lo_u32_b32 v{T}, -1, 0")
            A(f"v_mbcnt_hi_u32_b32 v{T}, -1, v{T}")
            A(f"v_lshrrev_b32 v{T}, 5, v{T}")
            A(f"v_lshlrev_b32 v{T}, 2, v{T}")                           # 4g
            A("s_nop 1")
            for qb in range(2):
                A(f"v_lshrrev_b32 v{T + 1}, 2, v{V_LSEOFF[qb]}")        # q
                A(f"v_add_u32 v{T + 1}, s{S_OFF}, v{T + 1}")
                A(f"v_sub_u32 v{T + 1}, v{T + 1}, v{T}")
                A(f"v_cvt_f32_i32 v{T + 1}, v{T + 1}")
                A(f"v_fma_f32 v{V_LSE2[qb]}, v{T + 1}, v{V_SL}, v{V_LSE2[qb]}")
                A(f"v_mov_b32 v{V_LP[qb]}, v{V_LSE2[qb]}")
            A(f"v_mov_b32 v{V_N0F}, 0")
        # ---- state: virtual fragments / scores / gradients 0 (these registers held dO / O and the prologue's addresses)
        for qb in range(2):
            for r in range(16):
                A(f"v_mov_b32 v{V_S[qb] + r}, 0")
                A(f"v_mov_b32 v{V_DP[qb] + r}, 0")
            for r in range(8):
                A(f"v_mov_b32 v{V_DS[qb] + r}, 0")
        for r in range(96):
            A(f"v_mov_b32 v{V_KR + r}, 0")
        A("s_waitcnt vmcnt(0)")
        # ---- q-block 0 runs half an iteration ahead: S0 / dP0 of the first stage (ring slot 1) before the loop, so that
        # the loop starts at j = n_min - 1 (no virtual iteration in front)
        A("s_barrier")                                            # every wave's pieces of the first stages have landed
        self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
        self.reset_dq(mfma_age=0)
        for ks in range(8):
            for which in ("kr", "vr"):
                for ins in self.frag_reads(which, ks, 1):
                    self.emit(ins)
        for _, mf in self.sdp_mfmas(0):
            self.emit(mf)
        self.drain_lds()
        L += self.out
        stamp(1)
        # ---- the loop: four copies (ring slots as immediates)
        report = {}
        for c in range(NRING):
            A(f"L_it{c}_%=:")
            if "bar" not in self.ko:
                A("s_barrier")
            # a wave without visible keys in this iteration's stages only moves data
            A(f"s_cmp_ge_i32 s{S_J}, s{S_WHI}")
            A(f"s_cbranch_scc1 L_data{c}_%=")
            A(f"s_add_u32 s{t}, s{S_J}, 2")
            A(f"s_cmp_lt_i32 s{t}, s{S_WLO}")
            A(f"s_cbranch_scc1 L_data{c}_%=")
            self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
            self.gen_iteration(c, cfg)
            report[c] = (dict(self.stats), len(self.out))
            L += self.out
            A(f"s_branch L_tail{c}_%=")
            A(f"L_data{c}_%=:")
            self.out = []
            self.gen_data_only(c)
            L += self.out
            A(f"L_tail{c}_%=:")
            A(f"s_add_u32 s{S_KSO}, s{S_KSO}, s{S_KSTG}")
            A(f"s_add_u32 s{S_VSO}, s{S_VSO}, s{S_VSTG}")
            if "vmwait" not in self.ko:
                A("s_waitcnt vmcnt(4)")
            A(f"s_add_u32 s{S_J}, s{S_J}, 1")
            A(f"s_add_u32 s{S_N0}, s{S_N0}, 32")
            A(f"s_cmp_lt_i32 s{S_J}, s{S_NMAX}")
            if c < NRING - 1:
                A("s_cbranch_scc0 L_done_%=")
            else:
                A("s_cbranch_scc1 L_it0_%=")
        A("L_done_%=:")
        A("s_waitcnt vmcnt(0) lgkmcnt(0)")
        stamp(2)
        A("s_barrier")                                            # every wave is done with the stage ring
        # ---- epilogue: dQ * softmax_scale -> 16 bit; through a wave-private LDS image ([64 rows][256 B + 16]) so that the
        # stores cover whole 256-byte rows (4 rows per instruction)
        E = V_KR
        lane, wbase, rbase, goff = E, E + 1, E + 2, E + 3
        A(f"v_mbcnt_lo_u32_b32 v{lane}, -1, 0")
        A(f"v_mbcnt_hi_u32_b32 v{lane}, -1, v{lane}")
        A(f"v_and_b32 v{E + 4}, 31, v{lane}")                      # q row of the accumulator columns
        A(f"v_lshrrev_b32 v{E + 5}, 5, v{lane}")                   # g
        A(f"v_lshrrev_b32 v{E + 6}, 4, v{lane}")                   # row of the lane's 16-byte chunk
        A(f"v_and_b32 v{E + 7}, 15, v{lane}")
        A(f"v_lshlrev_b32 v{E + 7}, 4, v{E + 7}")                  # its column byte
        A(f"s_mul_i32 s{t}, s{S_W1024}, {2 * EP_QB // 1024}")      # wave * 17408
        A(f"v_mul_u32_u24 v{wbase}, {EP_PITCH}, v{E + 4}")
        A(f"v_lshl_add_u32 v{wbase}, v{E + 5}, 3, v{wbase}")
        A(f"v_add_u32 v{wbase}, s{t}, v{wbase}")                   # write base: row * pitch + 8 g
        A(f"v_mul_u32_u24 v{rbase}, {EP_PITCH}, v{E + 6}")
        A(f"v_add3_u32 v{rbase}, v{rbase}, v{E + 7}, s{t}")        # read base: row * pitch + column byte
        A(f"v_readfirstlane_b32 s{t + 1}, v{V_DQRB}")
        A(f"v_readfirstlane_b32 s{t + 2}, v{V_R0}")
        A("s_nop 4")
        A(f"v_add_u32 v{E + 6}, s{t + 2}, v{E + 6}")               # global row
        A(f"v_mad_u32_u24 v{goff}, v{E + 6}, s{t + 1}, v{E + 7}")  # byte offset of the lane's chunk
        A(f"s_lshl_b32 s{t + 1}, s{t + 1}, 2")                     # 4 rows further
        A("s_nop 7")
        TT = V_T
        PK = V_VR
        for qb in range(2):
            for d in range(4):
                for r4 in range(4):
                    base = A_DQ[qb] + 16 * d + 4 * r4
                    tt = TT + 4 * (r4 & 1)
                    for e in range(4):
                        A(f"v_accvgpr_read_b32 v{tt + e}, a{base + e}")
                    for e in range(4):
                        A(f"v_mul_f32 v{tt + e}, s{S_SCALE}, v{tt + e}")
                    pk = PK + 2 * ((4 * d + r4) % 8)
                    A(f"{self.cvt} v{pk}, v{tt}, v{tt + 1}")
                    A(f"{self.cvt} v{pk + 1}, v{tt + 2}, v{tt + 3}")
                    A(f"ds_write_b64 v{wbase}, {vr(pk, 2)} offset:{qb * EP_QB + 64 * d + 16 * r4}")
        A("s_waitcnt lgkmcnt(0)")
        A(f"s_mov_b32 s{t + 3}, 0")
        ST = V_S[0]
        for qb in range(2):
            for j in range(8):
                A(f"ds_read_b128 {vr(ST + 4 * j, 4)}, v{rbase} offset:{qb * EP_QB + 4 * EP_PITCH * j}")
            for j in range(8):
                A(f"s_waitcnt lgkmcnt({7 - j})")
                A(f"buffer_store_dwordx4 {vr(ST + 4 * j, 4)}, v{goff}, {sr(S_DQRS, 4)}, s{t + 3} offen")
                A(f"s_add_u32 s{t + 3}, s{t + 3}, s{t + 1}")
            A("s_nop 1")
        # (no wait for the stores: their data left the registers at issue, and nothing below reads what they write)
        if timers:
            A("s_waitcnt vmcnt(0)")
            stamp(3)
            A("s_mov_b64 exec, 1")
            for i in range(4):
                A(f"buffer_store_dword v{V_TM + i}, v{V_LSEOFF[0]}, {sr(S_SDRS, 4)}, 0 offen offset:{4 * i}")
            A("s_mov_b64 exec, -1")
            A("s_waitcnt vmcnt(0)")
        A("s_branch L_end_%=")
        for qb in (0, 1):
            A(f"L_mask{qb}_%=:")
            L += self.gen_mask_routine(qb)
        A("L_end_%=:")
        return L, report


DEFAULT_CFG = {
    # MFMA index within phase B after which one DMA piece (M0 write + LDS-DMA) is emitted
    "dma_gaps": {2: [4, 9, 14, 19]},
    "lds_per_gap": 1,
}


def clobbers(alibi=False):
    c = ["memory", "vcc", "scc", "m0"]
    c += [f"v{i}" for i in range(V_ORB + 1 + (1 if alibi else 0), 256)]
    c += [f"a{i}" for i in range(256)]
    c += [f"s{i}" for i in range(S_J, S_LAST + 1)]
    return c


def main():
    cfg = dict(DEFAULT_CFG)
    ko = frozenset()
    for a in sys.argv[1:]:
        if a.startswith("--ko="):
            ko = frozenset(x for x in a[5:].split(",") if x)
        elif a.startswith("--cfg="):
            import json
            cfg.update(json.loads(a[6:]))
    if "dma_gaps" in cfg:
        cfg["dma_gaps"] = {int(k): v for k, v in cfg["dma_gaps"].items()}
    print("// GENERATED by gen_bwd_dq_asm.py - do not edit.  See that script for the schedule and the register map.")
    print("#pragma once")
    print(f"#define FA_BWD_DQ_ASM_LDS_BYTES {LDS_TOTAL}")
    for alibi in (False, True):
        tag = "ALIBI_" if alibi else ""
        for dt in ("bf16", "f16"):
            g = DQ(dt, alibi=alibi)
            g.ko = ko
            body, report = g.gen_body(cfg)
            print(f"#define FA_BWD_DQ_ASM_{tag}BODY_{dt.upper()} \\")
            for ln in body:
                print(f'    "{ln}\\n" \\')
            print('    ""')
            for k, (st, n) in report.items():
                print(f"// {dt} {tag}copy {k}: {n} lines, nop states {st['nop_states']}, lgkmcnt waits {st['lgkm_waits']}")
        cl = ", ".join(f'"{c}"' for c in clobbers(alibi))
        print(f"#define FA_BWD_DQ_ASM_{tag}CLOBBERS {cl}")

if __name__ == "__main__":
    main()
