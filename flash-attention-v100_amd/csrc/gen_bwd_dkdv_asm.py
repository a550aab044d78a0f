#!/usr/bin/env python3
"""Generator for the hand-scheduled gfx950 dK/dV kernel (fa_bwd_asm.hip): D = 128, no bias / dropout, dense.

Replaces the hot loop of the reference's kernel/fused_mha_backward.cu:367-474 (dK / dV accumulation) for
BASELINE config 2's shape family.  hipcc's version of this loop (fa_bwd_dkdv2_kernel) carries ~10 VALU
instructions per (q, key) element - twice what the arithmetic needs - and 1.75 LDS reads per MFMA; the kernel is
instruction-issue bound, so the stream is written out by hand:

  * workgroup = 4 waves (one per SIMD, the whole 512-register file each) on 128 keys, 32 keys per wave;
    dK^T / dV^T accumulators a[0:127], the wave's K / V fragments a[128:191] (B operands of S = Q K^T, dP = dO V^T);
  * a "stage" is 32 query rows of one q-head: Q and dO tiles arrive by LDS-DMA in a ring of six 16 KiB stages
    (both are read twice: by rows for S / dP, transposed for dK / dV), the row statistics (LSE log2e, D) of the
    stage as a 256-byte piece;
  * software pipeline over stages, unrolled by six (ring slot, buffer parity and every LDS address are immediates):
        iteration it:  MFMA  S, dP of stage it+1   and   dV, dK of stage it-1       (32 MFMAs)
                       VALU  P = exp2(S c - lse2), dS = P (dP - D), packing, of stage it   (5 per element)
                       DMA   stage it+3 (needed by iteration it+2: the end-of-iteration wait is a COUNTED vmcnt that
                             leaves this iteration's pieces in flight - memory latency is covered by a whole iteration)
    so the VALU work of a stage never depends on the MFMAs issued beside it; MFMA A operands come from two
    eight-entry fragment rings in a[192:255], read seven MFMAs ahead (the first seven transposed reads of an
    iteration are issued at the end of the previous one, across the barrier: their stage landed long ago);
  * the pipeline is filled and drained by VIRTUAL stages (fully masked, zero-filled LDS): one loop body, no variants;
  * masks (causal / window / key tail / rows past the sequence) set S = -inf in a called routine on edge stages only;
    the statistics are sanitised by the preprocess kernel (LSE = -inf -> +inf) so that -inf scores never meet NaN.

Run:  python gen_bwd_dkdv_asm.py > fa_bwd_asm_gen.h
"""
import sys
from gen_fwd_asm import Ins, Gen, rl, vr, ar, sr

# ------------------------------------------------------------------ LDS map
STG = 16384                    # one stage: Q tile 8 KiB + dO tile 8 KiB
NRING = 6
STATS = NRING * STG            # 6 x 256 B: [lse2 x 32 | D x 32] per ring slot
KST = 3 * STG                   # prologue: K tiles of the four waves (8 KiB each) staged in ring slots 3, 4 ...
VST = STATS + NRING * 256       # ... and the V tiles behind the statistics
LDS_TOTAL = VST + 4 * 8192
EP_PITCH = 272                  # epilogue: per-wave [32 keys][256 B + 16] images of dK and dV (bank-conflict-free 8-byte writes)
EP_T = 32 * EP_PITCH            # one tensor
EP_WAVE = 2 * EP_T              # one wave

# ------------------------------------------------------------------ SGPRs
S_QRS, S_DORS, S_KRS, S_VRS, S_DKRS, S_DVRS, S_STRS = 16, 20, 24, 28, 32, 36, 40     # buffer descriptors (s40: stats workspace)
S_C = 44                        # softmax_scale * log2e
S_SCALE = 45                    # softmax_scale (dK epilogue)
S_NITER = 46                    # real stages (group x tiles)
S_MT0, S_MT1 = 47, 48           # 32-row tile range of the key block
S_QROW32, S_DOROW32 = 49, 50    # bytes of 32 rows of q / dO
S_QHEAD, S_DOHEAD = 51, 52      # bytes between q heads in q / dO
S_STHEAD = 53                   # bytes between heads in the statistics planes (seqlen_q * 4)
S_W1024 = 54                    # wave * 1024
S_TL, S_NFULL = 55, 56          # mask predicate in 32-row tiles: a stage is unmasked iff (tile - TL) <u NFULL
S_FLO, S_FEND = 57, 58          # iterations [FLO, FEND) may run the fast copies: stage real and unmasked, DMA stage real,
                                # no head wrap (FLO >= FEND: never; the host sets that for GQA groups)
# (the q / dO / statistics descriptors start at the FIRST q-head of the kv-head's group)
# owned
S_IT = 60
S_T = 61                        # s61..s66 temps (s61: scratch; s64..s66: DMA source offsets of the iteration)
S_DQEND, S_DQS, S_DDOS, S_DSTS = 67, 68, 69, 70     # DMA stream: q soffset of the head's tile mt1; q / dO / stats soffsets
S_VMT = 71                      # VALU stream: tile index of stage `it`
S_OOB = 72
S_N0 = 73                       # q0 of the stage being masked
S_SUB, S_RET = 74, 76
S_MASKFN = (78, 80)
S_DIT = 82                      # DMA stream: stage index it + 3
# dS hand-off bodies (DKV(ds=True): the packed dS tile of every stage goes to a workspace, a one-GEMM dQ kernel reads it back -
# fa_bwd_dq_ds.hip): descriptor over the kv-head's group of q-heads, [head][32-key block][32-row tile][2 KiB]
S_DSRS = 84                     # in: s84..s87 buffer descriptor of the dS tiles (base: first q-head of the group)
S_DSWRAP = 88                   # in: what the tile offset advances by at a head wrap: head bytes - (mt1 - mt0 - 1) * 2048
S_DSN = 89                      # owned: tile offset of stage `it` (head + 32-row tile; the wave's key block rides in the lane offset)
S_DSO = 90                      # owned: ... of stage it - 1: the stage whose tile this iteration stores
S_DSX = 91                      # owned: S_DSO, or out of range for the virtual stages
S_LAST = 83
S_LAST_DS = 91

# ------------------------------------------------------------------ VGPRs
V_CBS = 16                      # in: v16..v19 = swizzled column byte of the lane's 16-byte DMA chunk for rows 4p + (lane >> 4), p = 0..3
V_KRB, V_VRB = 42, 43           # in (uniform): bytes per K / V row
V_DKRB, V_DVRB, V_KW0 = 208, 209, 210   # in (uniform): bytes per dK / dV row, first key of the wave
V_DSVO = 211                    # in (dS hand-off bodies): 16 lane + byte offset of the wave's 32-key block in a head's tiles
V_ROW = 20                      # in: 8 row-read addresses (swzt image, tile-relative)
V_TR = 28                       # in: 8 transposed-read addresses [h][d]
V_STB = 36                      # in: statistics read base (16 g)
V_DMAQ, V_DMADO, V_DMAST = 37, 38, 39
V_LOG, V_WID = 40, 41           # in: qlo - 4g (0x3fffffff: no row) and qhi - qlo of the lane's key
V_S = (64, 96)                  # S accumulators of the two buffer parities (16 each) ...
V_DP = (80, 112)                # ... and dP
V_P = (128, 144)                # packed P (8 regs) per parity
V_DS = (136, 152)               # packed dS
V_LSE = (160, 176)              # lse2 of the lane's 16 rows, per parity
V_D = 192                       # -D of the lane's 16 rows (stage it+1): srcC of the first dP MFMA
V_DMAQ2, V_DMADO2 = 224, 225    # owned: the lane's second piece (16 rows further)
V_HI = 226                      # owned: v20..v35 + 65536 (ring slots 4, 5 lie past the 16-bit offset field)
V_EP = 212                      # epilogue: eight 4-register entries v212..v243 (the loop's owned registers are dead by then)
V_TM = 244                      # measurement build: four time stamps
V_T = 44                        # temps v44..v63
# causal ALiBi variant (one q-head per kv-head): bias(q, key) = slope (key - off - q) in raw score units (sv = slope / softmax_scale)
# enters the S accumulator by ONE extra MFMA per stage (the compiler kernels' form, fa_common.h: alibi_pos_operand /
# alibi_lane_operand): A = row position and ones (lane constant), B = -sv, sv * key position (constant per pass) and the stage's
# tile term sv (kw0 - off - q0) split three ways (recomputed per stage: 10 VALU)
V_AA = 212                      # in: 4 regs, A operand (row position l31 in slots 0, 1; ones in slots 2 .. 6; lanes >= 32: zero)
V_AB01 = 216                    # in: 2 regs, B slots 0 .. 3: head / tail of -sv and of sv * l31 (lanes >= 32: zero)
V_SVL = 218                     # in: sv in lanes 0 .. 31, 0 in lanes 32 .. 63
V_AB = (220, 248)               # owned: B operand of the stage whose S is computed next, per buffer parity (4 regs each)
V_AX = 242                      # owned: two temps of the split
S_TT0 = 59                      # in: kw0 - off (integer): the stage's tile term is TT0 - 32 * tile
S_P32 = 59                      # in (bodies without ALiBi): != 0: the pass leaves an fp32 PARTIAL dK / dV - its dK / dV descriptors and row
                                # bytes describe [keys][kv-heads][128] fp32 rows of a split's slab (fa_bwd.hip: dkv_split_factor)
A_DK, A_DV, A_K, A_V = 0, 64, 128, 160
A_RR = 192                      # row-fragment ring (8 x 4)
A_TRR = 224                     # transposed-fragment ring (8 x 4)
NFR = 8


class DKV(Gen):
    def __init__(self, dtype, alibi=False, ds=False):
        Gen.__init__(self, dtype)
        self.alibi_kv = alibi
        self.ds = ds
        assert not (alibi and ds)

    def reset_dkv(self):
        self.now = 0
        self.last = {}
        self.lds_q = []
        self.srcc_rd = {}
        for par in (0, 1):
            for r in rl("v", V_S[par], 16) + rl("v", V_DP[par], 16):
                self.last[r] = (-8, "mfma", None)
        for r in rl("a", 0, 128):
            self.last[r] = (-8, "mfma", None)

    # ---- streams of one iteration (copy c of 4: ring slot and parity are compile-time) ----
    def sdp_stream(self, stage_slot, par):
        """S, dP of stage it+1: (reads, mfmas) - row fragments through a 4-entry ring"""
        items = []
        for ks in range(8):
            for (tens, acc, bfrag) in ((0, V_S[par], A_K), (1, V_DP[par], A_V)):
                n = ks * 2 + tens
                ring = A_RR + 4 * (n % NFR)
                off = stage_slot * STG + tens * 8192
                adr = V_ROW + ks
                if off >= 65536:
                    off, adr = off - 65536, V_HI + ks
                rd = Ins(f"ds_read_b128 {ar(ring, 4)}, v{adr} offset:{off}", "lds", [f"v{adr}"], rl("a", ring, 4))
                mf = self.mfma("v", acc, "a", ring, "a", bfrag + 4 * ks, ks == 0 and not (self.alibi_kv and tens == 0))
                if ks == 0 and tens == 1:          # dP = dO V^T - D: the accumulator starts from the -D rows
                    mf = Ins(f"{self.mf} {vr(acc, 16)}, {ar(ring, 4)}, {ar(bfrag, 4)}, {vr(V_D, 16)}", "mfma",
                             rl("a", ring, 4) + rl("a", bfrag, 4) + rl("v", V_D, 16), rl("v", acc, 16))
                items.append((rd, mf))
        if self.alibi_kv:           # S starts from the bias: one MFMA in front of the chain (no LDS operand)
            acc, b = V_S[par], V_AB[par]
            bias = Ins(f"{self.mf} {vr(acc, 16)}, {vr(V_AA, 4)}, {vr(b, 4)}, 0", "mfma", rl("v", V_AA, 4) + rl("v", b, 4), rl("v", acc, 16))
            items.insert(0, ([], bias))
        return items

    def alibi_valu(self, par):
        """B slots 4 .. 6 of the stage it + 1 (its S MFMAs run in this iteration): x = sv (TT0 - 32 tile) as head + middle + tail in
        the io type.  The tile index is mt0 + it + 1 (one q-head per kv-head: no wrap).  Lanes >= 32 compute zeros."""
        b, x, r = V_AB[par], V_AX, V_AX + 1
        t = S_T
        up = (lambda d, src: f"v_lshlrev_b32 v{d}, 16, v{src}") if self.dtype == "bf16" else (lambda d, src: f"v_cvt_f32_f16 v{d}, v{src}")
        V = lambda txt, rd, wr: Ins(txt, "valu", rd, wr)
        o = [Ins(f"s_add_u32 s{t}, s{S_MT0}, s{S_IT}", "salu", [], [f"s{t}", "scc"]),
             Ins(f"s_lshl_b32 s{t}, s{t}, 5", "salu", [f"s{t}"], [f"s{t}", "scc"]),
             Ins(f"s_sub_u32 s{t}, s{S_TT0}, s{t}", "salu", [f"s{t}"], [f"s{t}", "scc"]),
             Ins(f"s_sub_u32 s{t}, s{t}, 32", "salu", [f"s{t}"], [f"s{t}", "scc"]),
             V(f"v_cvt_f32_i32 v{x}, s{t}", [f"s{t}"], [f"v{x}"]),
             V(f"v_mul_f32 v{x}, v{x}, v{V_SVL}", [f"v{x}", f"v{V_SVL}"], [f"v{x}"]),
             V(f"{self.cvt} v{b + 2}, v{x}, 0", [f"v{x}"], [f"v{b + 2}"]),                # head (low half)
             V(up(r, b + 2), [f"v{b + 2}"], [f"v{r}"]),
             V(f"v_sub_f32 v{x}, v{x}, v{r}", [f"v{x}", f"v{r}"], [f"v{x}"]),           # x - head
             V(f"{self.cvt} v{b + 3}, v{x}, 0", [f"v{x}"], [f"v{b + 3}"]),                # middle
             V(up(b + 3, b + 3), [f"v{b + 3}"], [f"v{b + 3}"]),
             V(f"v_sub_f32 v{x}, v{x}, v{b + 3}", [f"v{x}", f"v{b + 3}"], [f"v{x}"]),   # tail
             V(f"{self.cvt} v{b + 2}, v{r}, v{b + 3}", [f"v{r}", f"v{b + 3}"], [f"v{b + 2}"]),   # slots 4, 5: head, middle (exact)
             V(f"{self.cvt} v{b + 3}, v{x}, 0", [f"v{x}"], [f"v{b + 3}"])]                # slot 6: tail, slot 7: 0
        return o

    def dvdk_stream(self, stage_slot, par):
        """dV, dK of stage it-1: transposed fragments through a 4-entry ring; B operands = packed P / dS"""
        items = []
        n = 0
        for t in range(2):
            for d in range(4):
                for (tens, acc, b) in ((1, A_DV + 16 * d, V_P[par] + 4 * t), (0, A_DK + 16 * d, V_DS[par] + 4 * t)):
                    ring = A_TRR + 4 * (n % NFR)
                    off = stage_slot * STG + tens * 8192 + t * 4096
                    hi = off >= 65536
                    off -= 65536 if hi else 0
                    adr = [(V_HI + 8 if hi else V_TR) + 4 * h + d for h in range(2)]
                    rds = [Ins(f"ds_read_b64_tr_b16 {ar(ring + 2 * h, 2)}, v{adr[h]} offset:{off}", "lds",
                               [f"v{adr[h]}"], rl("a", ring + 2 * h, 2)) for h in range(2)]
                    mf = self.mfma("a", acc, "a", ring, "v", b, False)
                    items.append((rds, mf))
                    n += 1
        return items

    def stats_reads(self, slot, par):
        """statistics of stage it+1: (-D reads -> V_D, consumed by this iteration's first dP MFMA;
        lse2 reads -> registers of the other parity, consumed by the VALU stream of the next iteration)"""
        dd, ll = [], []
        for i in range(4):
            for (which, base, out) in ((1, V_D, dd), (0, V_LSE[par], ll)):
                off = slot * 256 + which * 128 + i * 32          # the lane base (v36) carries STATS: past the 16-bit offset field
                out.append(Ins(f"ds_read_b128 {vr(base + 4 * i, 4)}, v{V_STB} offset:{off}", "lds", [f"v{V_STB}"], rl("v", base + 4 * i, 4)))
        return dd, ll

    def valu_stream(self, par):
        S, DP, L = V_S[par], V_DP[par], V_LSE[par]
        out = []
        for r in range(16 + 3):
            if r < 16:
                out.append(Ins(f"v_fma_f32 v{S + r}, v{S + r}, s{S_C}, -v{L + r}", "valu", [f"v{S + r}", f"v{L + r}"], [f"v{S + r}"]))
            if 0 <= r - 1 < 16:
                q = r - 1
                out.append(Ins(f"v_exp_f32 v{S + q}, v{S + q}", "trans", [f"v{S + q}"], [f"v{S + q}"]))
            if 0 <= r - 3 < 16:
                q = r - 3
                out.append(Ins(f"v_mul_f32 v{DP + q}, v{S + q}, v{DP + q}", "valu", [f"v{S + q}", f"v{DP + q}"], [f"v{DP + q}"]))
                if q % 2 == 1:
                    e = q // 2
                    out.append(Ins(f"{self.cvt} v{V_P[par] + e}, v{S + q - 1}, v{S + q}", "valu", [f"v{S + q - 1}", f"v{S + q}"], [f"v{V_P[par] + e}"]))
                    out.append(Ins(f"{self.cvt} v{V_DS[par] + e}, v{DP + q - 1}, v{DP + q}", "valu", [f"v{DP + q - 1}", f"v{DP + q}"], [f"v{V_DS[par] + e}"]))
        return out

    def dma_stream(self, slot, c, fast=False):
        """stage it+3 -> ring slot: 2 Q pieces + 2 dO pieces per wave (source offsets in s[S_T+3], s[S_T+4]; the second
        piece of a tile lies 16 rows further: its own lane offsets), then the 256-byte statistics piece by ONE wave
        (copy c: wave c % 4; offset s[S_T+5]).  -> list of (m0 write, DMA) pairs + the statistics block"""
        g = []
        base = slot * STG
        t = S_T
        sq, sdo, sst = (S_DQS, S_DDOS, S_DSTS) if fast else (t + 3, t + 4, t + 5)
        for (rs, so, vos, toff) in ((S_QRS, sq, (V_DMAQ, V_DMAQ2), 0), (S_DORS, sdo, (V_DMADO, V_DMADO2), 8192)):
            for jj in range(2):
                vo = vos[jj]
                g.append((Ins(f"s_add_u32 m0, s{S_W1024}, {base + toff + 4096 * jj}", "salu", [], ["m0", "scc"]),
                          Ins(f"buffer_load_dwordx4 v{vo}, {sr(rs, 4)}, s{so} offen lds", "dma", ["m0", f"v{vo}", f"s{so}"], [])))
        u = self.uid()
        st = [Ins(f"s_mov_b32 m0, {STATS + slot * 256}", "raw"), Ins(f"s_cmp_eq_u32 s{S_W1024}, {1024 * (c % 4)}", "raw"),
              Ins(f"s_cbranch_scc0 L_ns{u}_%=", "raw"),
              Ins(f"buffer_load_dword v{V_DMAST}, {sr(S_STRS, 4)}, s{sst} offen lds", "raw"), Ins(f"L_ns{u}_%=:", "raw")]
        return g, st

    def ds_stores(self, par, fast):
        """dS hand-off: the packed dS of stage it - 1 (B operands of this iteration's dK MFMAs, registers of the other parity) ->
        its 2-KiB tile: piece t = the lanes' four registers 4 t .. 4 t + 3, i.e. lane (key l31, g) leaves rows
        16 t + 4 g + (0..3) and 16 t + 8 + 4 g + (0..3) of its key as 16 contiguous bytes; two 1-KiB stores per wave.
        Virtual stages (it - 1 outside [0, n_iter)) store out of range (dropped by the descriptor's range check); the fast
        copies only run real stages (the host keeps FLO >= 1 for these bodies)."""
        so = S_DSO if fast else S_DSX
        if self.cfg_ds.get("ds_fixed"):             # timing experiment (wrong results): every stage lands on the head's first tile - no HBM stream
            so = "0"
            return [Ins(f"buffer_store_dwordx4 {vr(V_DS[par] + 4 * t, 4)}, v{V_DSVO}, {sr(S_DSRS, 4)}, 0 offen" + (f" offset:{1024 * t}" if t else ""),
                        "vmem", rl("v", V_DS[par] + 4 * t, 4) + [f"v{V_DSVO}"], []) for t in range(2)]
        extra = (" " + self.cfg_ds["ds_bits"].replace("+", " ")) if self.cfg_ds.get("ds_bits") else ""
        return [Ins(f"buffer_store_dwordx4 {vr(V_DS[par] + 4 * t, 4)}, v{V_DSVO}, {sr(S_DSRS, 4)}, s{so} offen" + (f" offset:{1024 * t}" if t else "") + extra,
                    "vmem", rl("v", V_DS[par] + 4 * t, 4) + [f"v{V_DSVO}"], []) for t in range(2)]

    def slots(self, c):
        """copy c (0..5): it = c - 1 (mod 6) at loop entry; stage s lives in ring slot (s + 1) % 6 and in the register
        buffers of parity s & 1 -> (slot of stage it-1, of stage it+1, of the DMA target it+3, parity of it, of it+-1)"""
        return (c + 5) % NRING, (c + 1) % NRING, (c + 3) % NRING, (c + 1) % 2, c % 2

    def pre_reads(self, c, pre):
        """the transposed reads of the first `pre` dV / dK MFMAs of copy c (issued at the end of the previous copy)"""
        sl_prev, _, _, _, par_oth = self.slots(c)
        out = []
        for (rds, mf) in self.dvdk_stream(sl_prev, par_oth)[:pre]:
            out += rds
        return out

    def wait_regs(self, regs):
        """one s_waitcnt covering every outstanding LDS read that writes one of `regs`"""
        if "lds" in self.ko:
            return
        touched = set(regs)
        idx = -1
        for i, q in enumerate(self.lds_q):
            if q & touched:
                idx = i
        if idx >= 0:
            n_after = len(self.lds_q) - 1 - idx
            if n_after < 15:
                self.out.append(f"s_waitcnt lgkmcnt({n_after})")
                self.now += 1
                self.stats["lgkm_waits"] += 1
            self.lds_q = self.lds_q[idx + 1:]

    def gen_iteration(self, c, cfg, carry, fast=False):
        """`carry`: the LDS queue the previous copy leaves behind (its lse2 reads and our first `pre` reads).
        fast: the host guarantees a real, mask-free stage, a real DMA stage and no head wrap - no predicate, no selects"""
        self.reset_dkv()
        self.lds_q = [set(x) for x in carry]
        sl_prev, sl_next, sl_dma, par_cur, par_oth = self.slots(c)
        A = self.raw
        t = S_T
        if not fast:
            self.gen_head(par_cur)
        self.gen_streams(c, cfg, fast)
        return [set(x) for x in self.lds_q]

    def gen_head(self, par_cur):
        A = self.raw
        t = S_T
        # ---- SALU head: is stage `it` masked at all?  virtual stages (it = -1, it >= n_iter: one unsigned compare)
        # and tiles outside [TL, TL + NFULL) call the mask routine ...
        u = self.uid()
        A(f"s_cmp_ge_u32 s{S_IT}, s{S_NITER}")
        A(f"s_cbranch_scc1 L_dm{u}_%=")
        A(f"s_sub_u32 s{t}, s{S_VMT}, s{S_TL}")
        A(f"s_cmp_lt_u32 s{t}, s{S_NFULL}")
        A(f"s_cbranch_scc1 L_nm{u}_%=")
        self.out.append(f"L_dm{u}_%=:")
        A(f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_MASKFN[par_cur], 2)}")
        self.out.append(f"L_nm{u}_%=:")
        # ... and the source offsets of the stage the DMA stream points at (it + 3): zeros past the last real stage
        A(f"s_cmp_lt_i32 s{S_DIT}, s{S_NITER}")
        A(f"s_cselect_b32 s{t + 3}, s{S_DQS}, s{S_OOB}")
        A(f"s_cselect_b32 s{t + 4}, s{S_DDOS}, s{S_OOB}")
        A(f"s_cselect_b32 s{t + 5}, s{S_DSTS}, s{S_OOB}")
        if self.ds:                                                   # stage it - 1 real?  (one unsigned compare: it - 1 <u n_iter)
            A(f"s_sub_u32 s{t}, s{S_IT}, 1")
            A(f"s_cmp_lt_u32 s{t}, s{S_NITER}")
            A(f"s_cselect_b32 s{S_DSX}, s{S_DSO}, s{S_OOB}")

    cfg_ds = {}

    def gen_streams(self, c, cfg, fast):
        self.cfg_ds = cfg
        sl_prev, sl_next, sl_dma, par_cur, par_oth = self.slots(c)
        # ---- streams
        sdp = self.sdp_stream(sl_next, par_oth)
        dvdk = self.dvdk_stream(sl_prev, par_oth)
        d_reads, l_reads = self.stats_reads(sl_next, par_oth)
        valu = (self.alibi_valu(par_oth) if self.alibi_kv else []) + self.valu_stream(par_cur)
        dma, dma_st = self.dma_stream(sl_dma, c, fast)
        # ---- interleave: 32 MFMAs = dV/dK of stage it-1 first (their operands are oldest), then S/dP of stage it+1
        mf_items = [("t", x) for x in dvdk] + [("r", x) for x in sdp]
        nv, vi = len(valu), 0
        pre = cfg.get("pre", 7)
        wb = cfg.get("wait_batch", 2)                                # one lgkmcnt wait per `wb` MFMAs
        assert pre < NFR and pre - (wb - 1) >= 3
        reads = []
        for kind, (rd, mf) in mf_items:
            reads.append(rd if isinstance(rd, list) else [rd])
        issued = pre                                              # (the previous copy issued them)
        dma_at = cfg.get("dma_at", [2, 8, 14, 20])
        st_at = cfg.get("st_at", 26)
        d_at = cfg.get("d_at", [1, 3, 5, 7])
        l_at = cfg.get("l_at", [19, 23, 27, 30])
        # (the two stores go out FIRST: the iteration's closing vmcnt(4) counts its four younger LOADS - loads return in order, so
        #  "at most 4 outstanding" proves the previous iteration's pieces have landed whatever the stores do (stores and loads
        #  complete out of order with respect to each other: a count that included them would prove nothing) - but it also
        #  waits for the stores themselves, which therefore get the whole iteration to be acknowledged)
        ds_at = cfg.get("ds_at", [0, 1]) if self.ds else []
        ds_st = self.ds_stores(par_oth, fast) if self.ds else []
        M = len(mf_items)
        for k, (kind, (rd, mf)) in enumerate(mf_items):
            if issued < M:
                for r in reads[issued]:
                    self.emit(r)
                issued += 1
            if k % wb == 0:                                           # operands of this MFMA and the next wb-1
                regs = []
                for kk in range(k, min(k + wb, M)):
                    regs += mf_items[kk][1][1].rd
                self.wait_regs([r for r in regs if not (r.startswith("v") and V_S[0] <= int(r[1:]) < V_P[0])])
            self.emit(mf)
            if (k + 1) in dma_at:
                self.emit(dma[dma_at.index(k + 1)][0])                # M0 one slot ahead of its DMA
            if k in dma_at:
                self.emit(dma[dma_at.index(k)][1])
            if k == st_at:
                for ins in dma_st:
                    self._emit_any(ins)
            if k in d_at:
                self.emit(d_reads[d_at.index(k)])
            if k in l_at:
                self.emit(l_reads[l_at.index(k)])
            if k in ds_at:
                self.emit(ds_st[ds_at.index(k)])
            take = -(-(nv - vi) // (M - k))
            for _ in range(take):
                if vi < nv:
                    self.emit(valu[vi])
                    vi += 1
        while vi < nv:
            self.emit(valu[vi])
            vi += 1
        for r in self.pre_reads((c + 1) % NRING, pre):
            self.emit(r)

    def _emit_any(self, ins):
        if ins.kind == "raw":
            if ins.txt.endswith(":"):
                self.out.append(ins.txt)
            else:
                self.raw(ins.txt)
        else:
            self.emit(ins)

    def salu_advance(self):
        """advance the VALU stream (stage it -> it+1) and the DMA stream (stage it+3 -> it+4): past the head's last
        tile both wrap to tile mt0 of the next q-head of the group"""
        o = []
        t = S_T
        o += [f"s_add_u32 s{S_VMT}, s{S_VMT}, 1",
              f"s_cmp_ge_i32 s{S_VMT}, s{S_MT1}",
              f"s_cselect_b32 s{S_VMT}, s{S_MT0}, s{S_VMT}"]
        if self.ds:                         # (scc still holds the wrap: next tile of the head, or tile mt0 of the next head)
            o += [f"s_mov_b32 s{S_DSO}, s{S_DSN}",
                  f"s_cselect_b32 s{t}, s{S_DSWRAP}, 2048",
                  f"s_add_u32 s{S_DSN}, s{S_DSN}, s{t}"]
        self.n_valu_adv = len(o)
        o += [f"s_add_u32 s{S_DIT}, s{S_DIT}, 1",
              f"s_add_u32 s{S_DQS}, s{S_DQS}, s{S_QROW32}",
              f"s_add_u32 s{S_DDOS}, s{S_DDOS}, s{S_DOROW32}",
              f"s_add_u32 s{S_DSTS}, s{S_DSTS}, 128",
              f"s_cmp_ge_u32 s{S_DQS}, s{S_DQEND}",
              f"s_cbranch_scc0 L_nw{{u}}_%=",
              # wrap: next q-head, first tile
              f"s_sub_u32 s{t}, s{S_MT1}, s{S_MT0}",
              f"s_mul_i32 s{t + 1}, s{t}, s{S_QROW32}",
              f"s_sub_u32 s{S_DQS}, s{S_DQS}, s{t + 1}",
              f"s_add_u32 s{S_DQS}, s{S_DQS}, s{S_QHEAD}",
              f"s_add_u32 s{S_DQEND}, s{S_DQEND}, s{S_QHEAD}",
              f"s_mul_i32 s{t + 1}, s{t}, s{S_DOROW32}",
              f"s_sub_u32 s{S_DDOS}, s{S_DDOS}, s{t + 1}",
              f"s_add_u32 s{S_DDOS}, s{S_DDOS}, s{S_DOHEAD}",
              f"s_lshl_b32 s{t + 1}, s{t}, 7",
              f"s_sub_u32 s{S_DSTS}, s{S_DSTS}, s{t + 1}",
              f"s_add_u32 s{S_DSTS}, s{S_DSTS}, s{S_STHEAD}",
              f"L_nw{{u}}_%=:"]
        return o

    def gen_mask_routine(self, par):
        o = []
        S, T = V_S[par], V_T
        tlo, tinf = f"v{T}", f"v{T + 1}"
        o.append(f"s_lshl_b32 s{S_N0}, s{S_VMT}, 5")                  # q0 of the stage
        o.append("s_nop 7")
        o.append("s_nop 3")
        o.append(f"v_subrev_u32 {tlo}, s{S_N0}, v{V_LOG}")           # lo_t = (qlo - 4g) - q0
        o.append(f"v_mov_b32 {tinf}, 0xff800000")
        # virtual stages: everything is masked
        o.append(f"s_cmp_ge_u32 s{S_IT}, s{S_NITER}")
        o.append(f"s_cbranch_scc0 L_mreg{par}_%=")
        o.append(f"v_mov_b32 {tlo}, 0x3fffffff")
        o.append(f"L_mreg{par}_%=:")
        o.append("s_nop 0")
        for r in range(16):
            c = (r & 3) + 8 * (r >> 2)
            t = f"v{T + 2 + (r & 3)}"
            o.append(f"v_sub_u32 {t}, {c}, {tlo}")
            o.append(f"v_cmp_gt_u32 vcc, {t}, v{V_WID}")
            o.append(f"v_cndmask_b32 v{S + r}, v{S + r}, {tinf}, vcc")
        o.append(f"s_setpc_b64 {sr(S_RET, 2)}")
        return o

    def gen_body(self, cfg):
        L = []
        A = L.append
        t = S_T
        pre = cfg.get("pre", 7)
        timers = cfg.get("timers", 0)          # measurement build: four s_memtime stamps land in the lane's dK row

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
        A("s_barrier")                                            # previous pass is done with LDS
        # ---- K / V tiles of this wave's 32 keys: eight 4-row LDS-DMA pieces each (whole 256-byte rows from memory)
        # into a wave-private staging area, read back as MFMA B fragments once everything has landed
        T = V_T
        A(f"v_mbcnt_lo_u32_b32 v{T}, -1, 0")
        A(f"v_mbcnt_hi_u32_b32 v{T}, -1, v{T}")                   # lane
        A(f"v_lshrrev_b32 v{T + 1}, 4, v{T}")                     # lane >> 4
        A(f"v_readfirstlane_b32 s{t}, v{V_KRB}")
        A(f"v_readfirstlane_b32 s{t + 1}, v{V_VRB}")
        A(f"v_readfirstlane_b32 s{t + 2}, v{V_KW0}")
        A("s_nop 4")
        for pp in range(4):
            A(f"v_add_u32 v{T + 10}, {4 * pp}, v{T + 1}")
            A(f"v_mad_u32_u24 v{T + 2 + pp}, v{T + 10}, s{t}, v{V_CBS + pp}")
            A(f"v_mad_u32_u24 v{T + 6 + pp}, v{T + 10}, s{t + 1}, v{V_CBS + pp}")
        A(f"s_mul_i32 s{t + 3}, s{t + 2}, s{t}")                  # byte offset of the wave's first key row
        A(f"s_mul_i32 s{t + 4}, s{t + 2}, s{t + 1}")
        A(f"s_lshl_b32 s{t}, s{t}, 4")
        A(f"s_lshl_b32 s{t + 1}, s{t + 1}, 4")
        A(f"s_add_u32 s{t}, s{t}, s{t + 3}")                      # ... and of its 16th
        A(f"s_add_u32 s{t + 1}, s{t + 1}, s{t + 4}")
        A(f"s_lshl_b32 s{t + 5}, s{S_W1024}, 3")                  # wave * 8192
        for (st0, vb, rs, so) in ((KST, T + 2, S_KRS, (t + 3, t)), (VST, T + 6, S_VRS, (t + 4, t + 1))):
            for pc in range(8):
                A(f"s_add_u32 m0, s{t + 5}, {st0 + 1024 * pc}")
                A("s_nop 0")
                A(f"buffer_load_dwordx4 v{vb + pc % 4}, {sr(rs, 4)}, s{so[pc // 4]} offen lds")
        # ---- lane offsets of the second piece of a tile (16 rows further)
        A(f"s_lshr_b32 s{t}, s{S_QROW32}, 1")
        A(f"v_add_u32 v{V_DMAQ2}, s{t}, v{V_DMAQ}")
        A(f"s_lshr_b32 s{t}, s{S_DOROW32}, 1")
        A(f"v_add_u32 v{V_DMADO2}, s{t}, v{V_DMADO}")
        for i in range(16):
            A(f"v_add_u32 v{V_HI + i}, 0x10000, v{V_ROW + i}")
        # ---- ring slots 5 and 0 (virtual stages -2, -1): zeros through an out-of-range source
        for slot in (NRING - 1, 0):
            for toff in (0, 8192):
                for jj in range(2):
                    A(f"s_add_u32 m0, s{S_W1024}, {slot * STG + toff + 4096 * jj}")
                    A("s_nop 0")
                    A(f"buffer_load_dwordx4 v{V_DMAQ}, {sr(S_QRS, 4)}, s{S_OOB} offen lds")
        A(f"s_mov_b32 s{S_DIT}, 0")
        A(f"s_mul_i32 s{S_DQEND}, s{S_MT1}, s{S_QROW32}")
        A(f"s_mul_i32 s{S_DQS}, s{S_MT0}, s{S_QROW32}")
        A(f"s_mul_i32 s{S_DDOS}, s{S_MT0}, s{S_DOROW32}")
        A(f"s_lshl_b32 s{S_DSTS}, s{S_MT0}, 7")

        def dma_advance():
            u = self.uid()
            return [x.replace("{u}", str(u)) for x in self.salu_advance()[self.n_valu_adv:]]      # (the DMA stream only)

        # ---- stages 0, 1 -> slots 1, 2 (stage 2 is the first iteration's DMA)
        for slot in (1, 2):
            A(f"s_cmp_lt_i32 s{S_DIT}, s{S_NITER}")
            A(f"s_cselect_b32 s{t + 3}, s{S_DQS}, s{S_OOB}")
            A(f"s_cselect_b32 s{t + 4}, s{S_DDOS}, s{S_OOB}")
            A(f"s_cselect_b32 s{t + 5}, s{S_DSTS}, s{S_OOB}")
            A("s_nop 3")
            self.out = []
            self.reset_dkv()
            pairs, st = self.dma_stream(slot, 0)                   # (statistics piece: wave 0)
            for (m0w, ld) in pairs:
                self.emit(m0w)
                self.emit(ld)
            for ins in st:
                self._emit_any(ins)
            L += self.out
            L += dma_advance()
        # ---- state
        for i in range(128):
            A(f"v_accvgpr_write_b32 a{i}, 0")
        for par in (0, 1):
            for r in range(16):
                A(f"v_mov_b32 v{V_S[par] + r}, 0")
                A(f"v_mov_b32 v{V_DP[par] + r}, 0")
            for r in range(8):
                A(f"v_mov_b32 v{V_P[par] + r}, 0")
                A(f"v_mov_b32 v{V_DS[par] + r}, 0")
            for r in range(16):
                A(f"v_mov_b32 v{V_LSE[par] + r}, 0")
        if self.alibi_kv:
            for par in (0, 1):
                A(f"v_mov_b32 v{V_AB[par]}, v{V_AB01}")
                A(f"v_mov_b32 v{V_AB[par] + 1}, v{V_AB01 + 1}")
                A(f"v_mov_b32 v{V_AB[par] + 2}, 0")
                A(f"v_mov_b32 v{V_AB[par] + 3}, 0")
        A(f"s_mov_b32 s{S_IT}, -1")
        A(f"s_mov_b32 s{S_VMT}, s{S_MT0}")
        A(f"s_sub_u32 s{S_VMT}, s{S_VMT}, 1")                      # stage -1 (virtual): advanced to mt0 before stage 0
        if self.ds:                                                # tile offsets of the stages -1 (never stored) and -2
            A(f"s_lshl_b32 s{S_DSN}, s{S_VMT}, 11")
            A(f"s_mov_b32 s{S_DSO}, s{S_OOB}")
        A("s_waitcnt vmcnt(0)")
        A(f"s_lshl_b32 s{t + 5}, s{S_W1024}, 3")
        A(f"s_add_u32 s{t + 4}, s{t + 5}, {VST}")
        for ks in range(8):
            A(f"v_add_u32 v{T + ks}, s{t + 5}, v{V_ROW + ks}")
            A(f"v_add_u32 v{T + 8 + ks}, s{t + 4}, v{V_ROW + ks}")
        for ks in range(8):
            A(f"ds_read_b128 {ar(A_K + 4 * ks, 4)}, v{T + ks} offset:{KST}")
            A(f"ds_read_b128 {ar(A_V + 4 * ks, 4)}, v{T + 8 + ks}")
        A("s_waitcnt lgkmcnt(0)")
        stamp(1)
        A("s_barrier")                                            # every wave's zero fill has landed, K / V staging is free
        ko_lds = "lds" in self.ko or "ldsv" in self.ko
        if not ko_lds:
            for r in self.pre_reads(0, pre):
                A(r.txt)
        # ---- the loop: six copies; the LDS queue a copy leaves behind seeds the next one (two generation passes)
        carries = {}
        for c in range(NRING):
            self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
            carries[c] = self.gen_iteration(c, cfg, [])
        report = {}
        use_fast = cfg.get("fast", True)
        for c in range(NRING):
            A(f"L_it{c}_%=:")
            if "bar" not in self.ko:
                A("s_barrier")
            if use_fast:                                            # FLO <= it < FEND: the mask-free copy of this slot phase
                A(f"s_cmp_ge_i32 s{S_IT}, s{S_FLO}")
                A(f"s_cbranch_scc0 L_gen{c}_%=")
                A(f"s_cmp_lt_i32 s{S_IT}, s{S_FEND}")
                A(f"s_cbranch_scc1 L_fb{c}_%=")
                A(f"L_gen{c}_%=:")
            self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
            left = self.gen_iteration(c, cfg, carries[(c - 1) % NRING])
            assert left == carries[c], "the carried LDS queue must not depend on its seed"
            report[c] = (dict(self.stats), len(self.out))
            L += self.out
            # ---- tail: advance both streams; the pieces of the PREVIOUS iteration must have landed before the next
            # barrier (this iteration's stay in flight: 4 per wave, 5 on the wave that also fetched the statistics)
            u = self.uid()
            for x in self.salu_advance():
                A(x.replace("{u}", str(u)))
            if "vmwait" not in self.ko:
                A(f"s_waitcnt vmcnt({cfg.get('ds_vmcnt', 4) if self.ds else 4})")      # (the wave that fetched the statistics, last, also waits for its first tile piece)
            A(f"s_add_u32 s{S_IT}, s{S_IT}, 1")
            A(f"s_cmp_le_i32 s{S_IT}, s{S_NITER}")
            if c < NRING - 1:
                A("s_cbranch_scc0 L_done_%=")
            else:
                A("s_cbranch_scc1 L_it0_%=")
                A("s_branch L_done_%=")
        if use_fast:
            # ---- fast copies: same slot phases; entered from a generic copy's head (after its barrier), left into the
            # next generic copy (which re-derives everything from `it`: VMT and DIT are restored on the way out)
            for c in range(NRING):
                A(f"L_f{c}_%=:")
                if "bar" not in self.ko:
                    A("s_barrier")
                A(f"L_fb{c}_%=:")
                self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
                left = self.gen_iteration(c, cfg, carries[(c - 1) % NRING], fast=True)
                assert left == carries[c]
                report[("fast", c)] = (dict(self.stats), len(self.out))
                L += self.out
                A(f"s_add_u32 s{S_DQS}, s{S_DQS}, s{S_QROW32}")
                A(f"s_add_u32 s{S_DDOS}, s{S_DDOS}, s{S_DOROW32}")
                A(f"s_add_u32 s{S_DSTS}, s{S_DSTS}, 128")
                if self.ds:
                    A(f"s_mov_b32 s{S_DSO}, s{S_DSN}")
                    A(f"s_add_u32 s{S_DSN}, s{S_DSN}, 2048")
                if "vmwait" not in self.ko:
                    A(f"s_waitcnt vmcnt({cfg.get('ds_vmcnt', 4) if self.ds else 4})")
                A(f"s_add_u32 s{S_IT}, s{S_IT}, 1")
                A(f"s_cmp_lt_i32 s{S_IT}, s{S_FEND}")
                A(f"s_cbranch_scc1 L_f{(c + 1) % NRING}_%=")
                A(f"s_add_u32 s{S_VMT}, s{S_MT0}, s{S_IT}")           # (no wrap inside the fast range: tile = mt0 + it)
                A(f"s_add_u32 s{S_DIT}, s{S_IT}, 3")
                A(f"s_branch L_it{(c + 1) % NRING}_%=")
        A("L_done_%=:")
        A("s_waitcnt vmcnt(0) lgkmcnt(0)")
        stamp(2)
        A("s_barrier")                                            # every wave is done with the stage ring
        ep0 = V_EP + (8 if self.alibi_kv else 0)      # (the ALiBi variant's INPUTS v212..v218 must survive: a workgroup may run two passes)
        # ---- epilogue: dK * softmax_scale, dV -> 16 bit; through a wave-private LDS image so that the stores cover
        # whole 256-byte rows (4 rows per instruction) instead of 8-byte shreds of 32 rows
        T = V_T
        A(f"v_mbcnt_lo_u32_b32 v{T}, -1, 0")
        A(f"v_mbcnt_hi_u32_b32 v{T}, -1, v{T}")                   # lane
        A(f"v_and_b32 v{T + 1}, 31, v{T}")                        # key of the accumulator columns
        A(f"v_lshrrev_b32 v{T + 2}, 5, v{T}")                     # g
        A(f"v_lshrrev_b32 v{T + 3}, 4, v{T}")                     # lane >> 4: row of the lane's 16-byte chunk
        A(f"v_and_b32 v{T + 4}, 15, v{T}")
        A(f"v_lshlrev_b32 v{T + 4}, 4, v{T + 4}")                 # its column byte
        A(f"s_mul_i32 s{t}, s{S_W1024}, {EP_WAVE // 1024}")       # wave * EP_WAVE
        A(f"v_mul_u32_u24 v{T + 5}, {EP_PITCH}, v{T + 1}")
        A(f"v_lshl_add_u32 v{T + 5}, v{T + 2}, 3, v{T + 5}")
        A(f"v_add_u32 v{T + 5}, s{t}, v{T + 5}")                  # write base: key * pitch + 8 g
        A(f"v_mul_u32_u24 v{T + 6}, {EP_PITCH}, v{T + 3}")
        A(f"v_add3_u32 v{T + 6}, v{T + 6}, v{T + 4}, s{t}")       # read base: row * pitch + column byte
        A(f"v_readfirstlane_b32 s{t + 1}, v{V_DKRB}")
        A(f"v_readfirstlane_b32 s{t + 2}, v{V_DVRB}")
        A(f"v_readfirstlane_b32 s{t + 3}, v{V_KW0}")
        A("s_nop 4")
        A(f"v_add_u32 v{T + 3}, s{t + 3}, v{T + 3}")              # global row
        A(f"v_mad_u32_u24 v{T + 7}, v{T + 3}, s{t + 1}, v{T + 4}")   # dK byte offset of the lane's chunk
        A(f"v_mad_u32_u24 v{T + 8}, v{T + 3}, s{t + 2}, v{T + 4}")   # dV
        A(f"s_lshl_b32 s{t + 1}, s{t + 1}, 2")                    # 4 rows further
        A(f"s_lshl_b32 s{t + 2}, s{t + 2}, 2")
        A("s_nop 7")
        if not self.alibi_kv:
            A(f"s_cmp_eq_u32 s{S_P32}, 0")
            A("s_cbranch_scc0 L_ep32_%=")
        for ti, (acc, scale) in enumerate(((A_DK, True), (A_DV, False))):
            for d in range(4):
                for r4 in range(4):
                    base = acc + 16 * d + 4 * r4
                    tt = T + 10 + 4 * (r4 & 1)
                    for e in range(4):
                        A(f"v_accvgpr_read_b32 v{tt + e}, a{base + e}")
                    if scale:
                        for e in range(4):
                            A(f"v_mul_f32 v{tt + e}, s{S_SCALE}, v{tt + e}")
                    pk = ep0 + 2 * ((4 * d + r4) % 8)
                    A(f"{self.cvt} v{pk}, v{tt}, v{tt + 1}")
                    A(f"{self.cvt} v{pk + 1}, v{tt + 2}, v{tt + 3}")
                    A(f"ds_write_b64 v{T + 5}, {vr(pk, 2)} offset:{ti * EP_T + 64 * d + 16 * r4}")
        A("s_waitcnt lgkmcnt(0)")
        for ti, (voff, rs, step) in enumerate(((T + 7, S_DKRS, t + 1), (T + 8, S_DVRS, t + 2))):
            A(f"s_mov_b32 s{t + 4}, 0")
            for j in range(8):
                A(f"ds_read_b128 {vr(ep0 + 4 * j, 4)}, v{T + 6} offset:{ti * EP_T + 4 * EP_PITCH * j}")
            for j in range(8):
                A(f"s_waitcnt lgkmcnt({7 - j})")
                A(f"buffer_store_dwordx4 {vr(ep0 + 4 * j, 4)}, v{voff}, {sr(rs, 4)}, s{t + 4} offen")
                A(f"s_add_u32 s{t + 4}, s{t + 4}, s{step}")
            A("s_nop 1")
        if timers:
            A("s_waitcnt vmcnt(0)")
            stamp(3)
            A("s_mov_b64 exec, 1")
            for i in range(4):
                A(f"buffer_store_dword v{V_TM + i}, v{T + 7}, {sr(S_DKRS, 4)}, 0 offen offset:{16 * i}")
            A("s_mov_b64 exec, -1")
            A("s_waitcnt vmcnt(0)")
        A("s_branch L_end_%=")
        if not self.alibi_kv:
            # ---- epilogue of a split pass: the fp32 accumulators as they are (dK scaled), 16 bytes per lane straight from the
            # registers: lane (key l31, g) holds columns 32 d + 8 r4 + 4 g .. + 3 of its key's row.  Rows past the sequence fall
            # outside the descriptor.  (32-byte runs per row - the epilogue is ~3 % of a pass and these rows stay in L2 for the
            # reduction kernel)
            A("L_ep32_%=:")
            A(f"v_add_u32 v{T + 1}, s{t + 3}, v{T + 1}")              # global key row of the lane's accumulator columns
            A(f"s_lshr_b32 s{t + 1}, s{t + 1}, 2")                    # (undo the "4 rows further" shifts)
            A(f"s_lshr_b32 s{t + 2}, s{t + 2}, 2")
            A(f"v_lshlrev_b32 v{T + 2}, 4, v{T + 2}")                 # 16 g
            A("s_nop 1")
            A(f"v_mad_u32_u24 v{T + 7}, v{T + 1}, s{t + 1}, v{T + 2}")   # dK byte offset: row * row bytes + 16 g
            A(f"v_mad_u32_u24 v{T + 8}, v{T + 1}, s{t + 2}, v{T + 2}")   # dV
            for ti, (acc, scale, voff, rs) in enumerate(((A_DK, True, T + 7, S_DKRS), (A_DV, False, T + 8, S_DVRS))):
                for d in range(4):
                    for r4 in range(4):
                        base = acc + 16 * d + 4 * r4
                        tt = ep0 + 4 * ((4 * d + r4) % 8)
                        for e in range(4):
                            A(f"v_accvgpr_read_b32 v{tt + e}, a{base + e}")
                        if scale:
                            for e in range(4):
                                A(f"v_mul_f32 v{tt + e}, s{S_SCALE}, v{tt + e}")
                        else:
                            A("s_nop 1")
                        A(f"buffer_store_dwordx4 {vr(tt, 4)}, v{voff}, {sr(rs, 4)}, 0 offen offset:{128 * d + 32 * r4}")
            A("s_branch L_end_%=")
        for par in (0, 1):
            A(f"L_mask{par}_%=:")
            L += self.gen_mask_routine(par)
        A("L_end_%=:")
        return L, report


def clobbers(alibi=False, ds=False):
    c = ["memory", "vcc", "scc", "m0"]
    keep = (V_DKRB, V_DVRB, V_KW0) + (tuple(range(V_AA, V_SVL + 1)) if alibi else ()) + ((V_DSVO,) if ds else ())
    c += [f"v{i}" for i in range(44, 256) if i not in keep]
    c += [f"a{i}" for i in range(256)]
    c += [f"s{i}" for i in range(S_IT, S_LAST + 1)]
    if ds:
        c += [f"s{i}" for i in range(S_DSN, S_LAST_DS + 1)]
    return c


def main():
    cfg = {}
    ko = frozenset()
    for a in sys.argv[1:]:
        if a.startswith("--ko="):
            ko = frozenset(x for x in a[5:].split(",") if x)
        elif a.startswith("--cfg="):
            import json
            cfg.update(json.loads(a[6:]))
    print("// GENERATED by gen_bwd_dkdv_asm.py - do not edit.  See that script for the schedule and the register map.")
    print("#pragma once")
    print(f"#define FA_BWD_ASM_LDS_BYTES {LDS_TOTAL}")
    print(f"#define FA_BWD_ASM_STATS_OFF {STATS}")
    for (alibi, ds) in ((False, False), (True, False), (False, True)):
        tag = "ALIBI_" if alibi else ("DS_" if ds else "")
        for dt in ("bf16", "f16"):
            g = DKV(dt, alibi=alibi, ds=ds)
            g.ko = ko
            body, report = g.gen_body(cfg)
            print(f"#define FA_BWD_DKDV_ASM_{tag}BODY_{dt.upper()} \\")
            for ln in body:
                print(f'    "{ln}\\n" \\')
            print('    ""')
            for k, (st, n) in report.items():
                print(f"// {dt} {tag}copy {k}: {n} lines, nop states {st['nop_states']}, lgkmcnt waits {st['lgkm_waits']}")
        cl = ", ".join(f'"{c}"' for c in clobbers(alibi, ds))
        print(f"#define FA_BWD_DKDV_ASM_{tag}CLOBBERS {cl}")


if __name__ == "__main__":
    main()
