#!/usr/bin/env python3
"""Generator for the warp-specialised gfx950 forward (fa_fwd_ws.hip): D = 128, no bias / dropout / paging.

Why: with one wave per SIMD (gen_fwd_asm.py) every LDS read (~10 cycles) and every LDS-DMA issue (~64 cycles) of a
wave is serial with its own MFMAs - measured by knock-outs, independent of where they are placed.  Here a SIMD hosts
TWO waves with different jobs, so the loads of one wave run beside the matrix instructions of the other:

  S wave (waves 0-3): S^T = K Q^T for its 64 query rows (32 MFMAs / 64-key tile), masks, online softmax, packs P to
                      16 bit and hands it to its partner through LDS (8 KiB per tile, lane-private slots);
  O wave (waves 4-7): issues ALL LDS-DMA of the workgroup, reads P and the V^T fragments, O^T += V^T P^T (32 MFMAs).

The O wave runs one tile behind its S wave; one s_barrier per tile hands over P, publishes the DMA'd tiles and frees
ring slots (K, V and P rings are 2 deep, the loop bodies are unrolled by two so every LDS address is an immediate).
A rescale of the running maximum (rare: deferred, threshold 2^8) travels as a flag + alpha vector in the P slot.

Register budget: 256 per wave (2 waves per SIMD): arch v0..v175, acc a0..a79.
Replaces the hot loop of the reference's kernel/fused_mha_forward.cu:148-209.
"""
import sys
from gen_fwd_asm import Ins, Gen, rl, vr, ar, sr

# ------------------------------------------------------------------ LDS map
K_SLOT = (0, 16384)
V_SLOT = (32768, 49152)
P_BASE = 65536
P_SLOT_BYTES = 9728            # 8 KiB P + 512 B alpha + 1 KiB flags (a private 16-byte cell per lane)
P_ALPHA = 8192
P_FLAGS = 8704
LDS_TOTAL = P_BASE + 4 * 2 * P_SLOT_BYTES

# ------------------------------------------------------------------ SGPRs (inputs s16..s49, s81..s83; owned s50..s80)
S_QRS, S_KRS, S_VRS, S_ORS, S_LRS = 16, 20, 24, 28, 32
S_KTILE, S_VTILE, S_K16, S_V16, S_C = 36, 37, 38, 39, 40
S_IIN, S_NMAX, S_WLO, S_WHI = 41, 42, 43, 44
S_HIMIN, S_LOMAX = (45, 47), (46, 48)
S_W1024 = 49
S_RC, S_ROLE = 81, 82
S_I = 50
S_OOB = 51
S_T = 52                       # s52..s57 temps
S_N0 = 58
S_PSLOT = 59                   # byte offset of the current P slot (for the routines)
S_FLAG = (60, 61)              # S wave: q-block rescaled this tile
S_SUB, S_RET = 62, 64
S_MASKFN, S_RESCFN = (66, 68), (70, 72)
S_ORESC = 74                   # s[74:75] O-wave rescale routine
S_KJ, S_VJ = 76, 78            # unused spare
S_LAST = 80

# ------------------------------------------------------------------ S wave registers
SV_QOFF = (8, 9)
SV_LSEOFF = (10, 11)
SV_LOG, SV_WID = (12, 14), (13, 15)
SV_KBASE = 16                  # 8
V_PBASE, V_PALPHA = 24, 25     # both roles: pair P ring base + lane * 16 / + 8192 + lane * 4
SV_MRUN, SV_NEGM, SV_L, SV_L2, SV_THR = (26, 31), (27, 32), (28, 33), (29, 34), (30, 35)
SV_T = 36                      # 28 temps
SV_S = (64, 96)                # per q-block: kb0 16, kb1 16
SV_P = (128, 144)              # per q-block: 4 k-steps x 4
SV_KRING = [("a", 64), ("a", 68), ("a", 72), ("a", 76), ("v", 160), ("v", 164), ("v", 168), ("v", 172)]
SA_Q = (0, 32)
# ------------------------------------------------------------------ O wave registers
OV_OOFF = (8, 9)
OV_DMAK, OV_DMAV, OV_VBASE = 10, 11, 12
OV_P = 26                      # P fragments [qb][ks] x 4 = 32   (v24 / v25 are the P ring addresses)
OV_VRING = [58 + 4 * i for i in range(12)]      # v58..v105
OV_T = 154                     # temps v154..v173
O_ACC = [("a", 16 * i) for i in range(5)] + [("v", 106 + 16 * i) for i in range(3)]     # O^T[qb][d], index qb * 4 + d
N_ARCH, N_ACC = 176, 80


TM = 76                        # s76..s79: timer accumulators, s80: reference (cfg timers=1, measurement builds only)


class WS(Gen):
    timers = False

    def stamp(self, k):
        """accumulate (now - previous stamp) into timer k; k < 0: only set the reference"""
        if not self.timers:
            return
        self.drain_lds()
        self.raw(f"s_memtime {sr(S_SUB, 2)}")
        self.raw("s_waitcnt lgkmcnt(0)")
        if k >= 0:
            self.raw(f"s_sub_u32 s{S_N0}, s{S_SUB}, s{TM + 4}")
            self.raw(f"s_add_u32 s{TM + k}, s{TM + k}, s{S_N0}")
        self.raw(f"s_mov_b32 s{TM + 4}, s{S_SUB}")

    def stamp_lines(self, k):
        save = self.out
        self.out = []
        self.stamp(k)
        r, self.out = self.out, save
        return r

    def dump_timers(self, base_row, rowreg):
        o = []
        if not self.timers:
            return o
        o.append(f"v_readfirstlane_b32 s{S_N0}, v{rowreg}")
        for k in range(4):
            o.append(f"v_cvt_f32_u32 v{150 + k}, s{TM + k}")
        o.append(f"v_mov_b32 v149, {4 * base_row}")
        o.append(f"v_add_u32 v149, s{S_N0}, v149")
        for k in range(4):
            o.append(f"buffer_store_dword v{150 + k}, v149, {sr(S_LRS, 4)}, 0 offen offset:{4 * k}")
        o.append("s_waitcnt vmcnt(0)")
        return o

    def reset_ws(self, role):
        self.role = role.lower()
        self.now = 0
        self.last = {}
        self.lds_q = []
        self.srcc_rd = {}
        regs = []
        if role == "S":
            for qb in (0, 1):
                regs += rl("v", SV_S[qb], 32)
        else:
            for (p, b) in O_ACC:
                regs += rl(p, b, 16)
        for r in regs:
            self.last[r] = (-8, "mfma", None)

    def lds_write(self, txt, rd):
        return Ins(txt, "lds", rd, [])

    # ---------------------------------------------------------------- S wave
    def s_qk(self, parity, cfg):
        """32 MFMAs S^T[qb][kb] (+)= K[kb][ks] Q[qb][ks]^T with the K fragments streamed through a small ring."""
        ring = SV_KRING[:cfg.get("kring", 8)]
        order = [(kb, ks) for ks in range(8) for kb in range(2)]
        kslot = K_SLOT[parity]

        def kread(n):
            kb, ks = order[n]
            p, b = ring[n % len(ring)]
            return Ins(f"ds_read_b128 {p}[{b}:{b + 3}], v{SV_KBASE + ks} offset:{kslot + kb * 8192}", "lds",
                       [f"v{SV_KBASE + ks}"], rl(p, b, 4))
        pre = len(ring) - 1
        for n in range(min(pre, 16)):
            self.emit(kread(n))
        nxt = min(pre, 16)
        for n, (kb, ks) in enumerate(order):
            p, b = ring[n % len(ring)]
            if nxt < 16:                               # the slot freed by fragment n-1 (its two MFMAs are issued)
                self.emit(kread(nxt))
                nxt += 1
            for qb in range(2):
                self.emit(self.mfma("v", SV_S[qb] + 16 * kb, p, b, "a", SA_Q[qb] + 4 * ks, ks == 0))

    def s_softmax(self, qb, parity):
        S, P, T = SV_S[qb], SV_P[qb], SV_T
        s = [f"v{S + i}" for i in range(32)]
        vals, tn = list(s), 0
        while len(vals) > 1:
            nxt, i = [], 0
            while i + 2 < len(vals):
                t = f"v{T + tn}"
                tn += 1
                self.emit(Ins(f"v_max3_f32 {t}, {vals[i]}, {vals[i + 1]}, {vals[i + 2]}", "valu", vals[i:i + 3], [t]))
                nxt.append(t)
                i += 3
            rest = vals[i:]
            if len(rest) == 2 and not nxt:
                t = f"v{T + tn}"
                tn += 1
                self.emit(Ins(f"v_max_f32 {t}, {rest[0]}, {rest[1]}", "valu", rest, [t]))
                nxt, rest = [t], []
            vals = nxt + rest
        mx = vals[0]
        ta = f"v{T + 20}"
        mxr = f"v{T + 23 + qb}"
        self.emit(Ins(f"v_mov_b32 {ta}, {mx}", "valu", [mx], [ta]))
        self.emit(Ins(f"v_permlane32_swap_b32 {ta}, {mx}", "swap", [ta, mx], [ta, mx]))
        self.emit(Ins(f"v_max_f32 {mxr}, {ta}, {mx}", "valu", [ta, mx], [mxr]))
        self.emit(Ins(f"v_cmp_lt_f32 vcc, v{SV_THR[qb]}, {mxr}", "valu", [mxr, f"v{SV_THR[qb]}"], ["vcc"]))
        u = self.uid()
        self.emit(Ins(f"s_cbranch_vccz L_nr{u}_%=", "salu", ["vcc"], []))
        self.raw(f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_RESCFN[qb], 2)}")
        self.out.append(f"L_nr{u}_%=:")
        negm, l0, l1 = f"v{SV_NEGM[qb]}", f"v{SV_L[qb]}", f"v{SV_L2[qb]}"
        pslot = P_SLOT_BYTES * parity
        for r0 in range(0, 32, 2):
            rs = (r0, r0 + 1)
            for r in rs:
                self.emit(Ins(f"v_fma_f32 {s[r]}, {s[r]}, s{S_C}, {negm}", "valu", [s[r], negm], [s[r]]))
            for r in rs:
                self.emit(Ins(f"v_exp_f32 {s[r]}, {s[r]}", "trans", [s[r]], [s[r]]))
            for r in rs:
                ll = l0 if r % 2 == 0 else l1
                self.emit(Ins(f"v_add_f32 {ll}, {ll}, {s[r]}", "valu", [ll, s[r]], [ll]))
            r = r0 + 1
            kb, rr = r // 16, r % 16
            ks, e = 2 * kb + rr // 8, (rr % 8) // 2
            dst = f"v{P + 4 * ks + e}"
            self.emit(Ins(f"{self.cvt} {dst}, {s[r - 1]}, {s[r]}", "valu", [s[r - 1], s[r]], [dst]))
            if e == 3:                                 # k-step complete: hand it to the O wave
                off = pslot + (qb * 4 + ks) * 1024
                self.emit(self.lds_write(f"ds_write_b128 v{V_PBASE}, {vr(P + 4 * ks, 4)} offset:{off}",
                                         [f"v{V_PBASE}"] + rl("v", P + 4 * ks, 4)))

    def s_mask_check(self, qb):
        t, u = S_T, self.uid()
        for l in (f"s_lshl_b32 s{S_N0}, s{S_I}, 6",
                  f"s_add_u32 s{t}, s{S_N0}, 63",
                  f"s_cmp_gt_i32 s{t}, s{S_HIMIN[qb]}",
                  f"s_cbranch_scc1 L_dm{u}_%=",
                  f"s_cmp_lt_i32 s{S_N0}, s{S_LOMAX[qb]}",
                  f"s_cbranch_scc0 L_nm{u}_%="):
            self.raw(l)
        self.out.append(f"L_dm{u}_%=:")
        self.raw("s_nop 7")                             # the S accumulators may still be in flight
        self.raw("s_nop 3")
        self.raw(f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_MASKFN[qb], 2)}")
        self.out.append(f"L_nm{u}_%=:")

    def s_iteration(self, parity, cfg):
        self.reset_ws("S")
        self.raw(f"s_mov_b32 s{S_PSLOT}, {P_SLOT_BYTES * parity}")
        self.stamp(0)                                   # 0: barrier wait + dispatch
        self.s_qk(parity, cfg)
        self.stamp(1)                                   # 1: QK phase
        for qb in range(2):
            self.s_mask_check(qb)
            self.s_softmax(qb, parity)
        self.stamp(2 )                                  # 2: softmax + P writes
        # rescale flags of this tile
        f0, f1 = f"v{SV_T}", f"v{SV_T + 1}"
        self.emit(Ins(f"v_mov_b32 {f0}, s{S_FLAG[0]}", "valu", [], [f0]))
        self.emit(Ins(f"v_mov_b32 {f1}, s{S_FLAG[1]}", "valu", [], [f1]))
        self.emit(self.lds_write(f"ds_write_b64 v{V_PBASE}, {vr(SV_T, 2)} offset:{P_SLOT_BYTES * parity + P_FLAGS}",
                                 [f"v{V_PBASE}", f0, f1]))
        self.raw(f"s_mov_b64 {sr(S_FLAG[0], 2)}, 0")

    def s_mask_routine(self, qb):
        o, S, T = [], SV_S[qb], SV_T
        tlo, tinf = f"v{T}", f"v{T + 1}"
        o.append(f"v_subrev_u32 {tlo}, s{S_N0}, v{SV_LOG[qb]}")
        o.append(f"v_mov_b32 {tinf}, 0xff800000")
        o.append("s_nop 0")
        for kb in range(2):
            for r in range(16):
                c = 32 * kb + (r & 3) + 8 * (r >> 2)
                t = f"v{T + 2 + (r & 3)}"
                o.append(f"v_sub_u32 {t}, {c}, {tlo}")
                o.append(f"v_cmp_gt_u32 vcc, {t}, v{SV_WID[qb]}")
                o.append(f"v_cndmask_b32 v{S + 16 * kb + r}, v{S + 16 * kb + r}, {tinf}, vcc")
        o.append(f"s_setpc_b64 {sr(S_RET, 2)}")
        return o

    def s_rescale_routine(self, qb):
        o, T = [], SV_T + 8
        mxr = f"v{SV_T + 23 + qb}"
        mrun, negm, l, l2, thr = (f"v{x[qb]}" for x in (SV_MRUN, SV_NEGM, SV_L, SV_L2, SV_THR))
        mxs, mnew, muse, al, ad = (f"v{T + i}" for i in range(5))
        o += [f"v_mul_f32 {mxs}, s{S_C}, {mxr}",
              f"v_max_f32 {mnew}, {mrun}, {mxs}",
              f"v_max_f32 {muse}, 0xff7fffff, {mnew}",
              f"v_sub_f32 {al}, {mrun}, {muse}",
              f"v_exp_f32 {al}, {al}",
              f"v_mov_b32 {mrun}, {mnew}",
              f"v_sub_f32 {negm}, 0, {muse}",
              f"v_mul_f32 {l}, {l}, {al}",
              f"v_mul_f32 {l2}, {l2}, {al}",
              f"v_add_f32 {thr}, 0x41000000, {mnew}",
              f"v_mul_f32 {thr}, s{S_RC}, {thr}",
              f"v_add_u32 {ad}, s{S_PSLOT}, v{V_PALPHA}",
              f"ds_write_b32 {ad}, {al} offset:{256 * qb}",
              f"s_mov_b32 s{S_FLAG[qb]}, 1",
              f"s_setpc_b64 {sr(S_RET, 2)}"]
        return o

    # ---------------------------------------------------------------- O wave
    def o_dma_groups(self, iparity):
        """K(i+1) -> K slot (i+1)&1, V(i) -> V slot i&1 ; tile offsets were put into s[S_T+2], s[S_T+3] by o_iteration."""
        g = []
        kso, vso = S_T + 2, S_T + 3
        for (rs, so, s16, vo, base) in ((S_KRS, kso, S_K16, OV_DMAK, K_SLOT[1 - iparity]), (S_VRS, vso, S_V16, OV_DMAV, V_SLOT[iparity])):
            for jj in range(4):
                p = [Ins(f"s_add_u32 m0, s{S_W1024}, {base + 4096 * jj}", "salu", [], ["m0", "scc"]),
                     Ins(f"buffer_load_dwordx4 v{vo}, {sr(rs, 4)}, s{so} offen lds", "dma", ["m0", f"v{vo}", f"s{so}"], [])]
                if jj < 3:
                    p.append(Ins(f"s_add_u32 s{so}, s{so}, s{s16}", "salu", [f"s{so}"], [f"s{so}", "scc"]))
                g.append(p)
        return g

    def o_tile_offsets(self):
        t = S_T
        for l in (f"s_add_u32 s{t}, s{S_I}, 1",
                  f"s_mul_i32 s{t + 2}, s{t}, s{S_KTILE}",
                  f"s_cmp_lt_i32 s{t}, s{S_NMAX}",
                  f"s_cselect_b32 s{t + 2}, s{t + 2}, s{S_OOB}",
                  f"s_mul_i32 s{t + 3}, s{S_I}, s{S_VTILE}",
                  f"s_cmp_lt_i32 s{S_I}, s{S_NMAX}",
                  f"s_cselect_b32 s{t + 3}, s{t + 3}, s{S_OOB}"):
            self.raw(l)

    def o_iteration(self, iparity, active, cfg):
        """iteration i (parity iparity): DMA for K(i+1), V(i); if active, PV of tile i-1 (parity 1 - iparity)."""
        self.reset_ws("O")
        tp = 1 - iparity
        self.o_tile_offsets()
        dma = self.o_dma_groups(iparity)
        if not active:
            for grp in dma:
                for ins in grp:
                    self.emit(ins)
            return
        pslot = P_SLOT_BYTES * tp
        self.stamp(0)
        # ---- rescale flags of tile i-1
        f = OV_T
        self.emit(Ins(f"ds_read_b64 {vr(f, 2)}, v{V_PBASE} offset:{pslot + P_FLAGS}", "lds", [f"v{V_PBASE}"], rl("v", f, 2)))
        # ---- P fragments
        for qb in range(2):
            for ks in range(4):
                b = OV_P + (qb * 4 + ks) * 4
                self.emit(Ins(f"ds_read_b128 {vr(b, 4)}, v{V_PBASE} offset:{pslot + (qb * 4 + ks) * 1024}", "lds",
                              [f"v{V_PBASE}"], rl("v", b, 4)))
        # ---- V fragment ring; the DMA of the next tiles is issued between the first reads, i.e. while the partner
        # S wave owns the matrix pipe (its QK phase opens the iteration)
        ring = OV_VRING[:cfg.get("vring", 12)]
        order = [(d, ks) for ks in range(4) for d in range(4)]
        vslot = V_SLOT[tp]

        def vread(n):
            d, ks = order[n]
            b = ring[n % len(ring)]
            off = vslot + ks * 4096 + d * 256
            return [Ins(f"ds_read_b64_tr_b16 {vr(b, 2)}, v{OV_VBASE} offset:{off}", "lds", [f"v{OV_VBASE}"], rl("v", b, 2)),
                    Ins(f"ds_read_b64_tr_b16 {vr(b + 2, 2)}, v{OV_VBASE} offset:{off + 2048}", "lds", [f"v{OV_VBASE}"], rl("v", b + 2, 2))]
        pre = min(len(ring) - 1, 16)
        dma_first = cfg.get("o_dma_first", True)
        for n in range(pre):
            for ins in vread(n):
                self.emit(ins)
            if dma_first and dma:
                for ins in dma.pop(0):
                    self.emit(ins)
            if n == 2:
                # flags -> scalar, rare call (the flag read is the oldest LDS operation in flight)
                self.emit(Ins(f"v_readfirstlane_b32 s{S_T + 4}, v{f}", "valu", [f"v{f}"], [f"s{S_T + 4}"]))
                self.emit(Ins(f"v_readfirstlane_b32 s{S_T + 5}, v{f + 1}", "valu", [f"v{f + 1}"], [f"s{S_T + 5}"]))
                u = self.uid()
                self.raw(f"s_mov_b32 s{S_PSLOT}, {pslot}")
                self.raw(f"s_or_b32 s{S_T}, s{S_T + 4}, s{S_T + 5}")
                self.raw(f"s_cmp_eq_u32 s{S_T}, 0")
                self.raw(f"s_cbranch_scc1 L_onr{u}_%=")
                self.raw(f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_ORESC, 2)}")
                self.out.append(f"L_onr{u}_%=:")
        nxt = pre
        self.stamp(1)                                   # 1: P / V reads + DMA issue
        dma_at = cfg.get("o_dma_at", [1, 3, 5, 7, 9, 11, 13, 15])
        for n, (d, ks) in enumerate(order):
            b = ring[n % len(ring)]
            if nxt < 16:
                for ins in vread(nxt):
                    self.emit(ins)
                nxt += 1
            for qb in range(2):
                p, ob = O_ACC[qb * 4 + d]
                self.emit(self.mfma(p, ob, "v", b, "v", OV_P + (qb * 4 + ks) * 4, False))
            for _ in range(dma_at.count(n)):
                if dma:
                    for ins in dma.pop(0):
                        self.emit(ins)
        for grp in dma:
            for ins in grp:
                self.emit(ins)
        self.stamp(2)                                   # 2: PV phase

    def o_rescale_routine(self):
        """flags in s[S_T+4], s[S_T+5]; O^T[qb] *= alpha[row] for the flagged q-blocks."""
        o = ["s_nop 7", "s_nop 7"]
        T = OV_T + 4
        ad, al = f"v{T}", f"v{T + 1}"
        for qb in range(2):
            o.append(f"s_cmp_eq_u32 s{S_T + 4 + qb}, 0")
            o.append(f"s_cbranch_scc1 L_ors{qb}_%=")
            o.append(f"v_add_u32 {ad}, s{S_PSLOT}, v{V_PALPHA}")
            o.append(f"ds_read_b32 {al}, {ad} offset:{256 * qb}")
            o.append("s_waitcnt lgkmcnt(0)")
            for d in range(4):
                p, b = O_ACC[qb * 4 + d]
                for i in range(0, 16, 4):
                    if p == "a":
                        for e in range(4):
                            o.append(f"v_accvgpr_read_b32 v{T + 2 + e}, a{b + i + e}")
                        for e in range(4):
                            o.append(f"v_mul_f32 v{T + 2 + e}, v{T + 2 + e}, {al}")
                        for e in range(4):
                            o.append(f"v_accvgpr_write_b32 a{b + i + e}, v{T + 2 + e}")
                    else:
                        for e in range(4):
                            o.append(f"v_mul_f32 v{b + i + e}, v{b + i + e}, {al}")
            o.append(f"L_ors{qb}_%=:")
        o.append("s_nop 1")
        o.append(f"s_setpc_b64 {sr(S_RET, 2)}")
        return o

    # ---------------------------------------------------------------- body
    def gen_body(self, cfg):
        L = []
        A = L.append
        A("s_nop 7")
        A(f"s_getpc_b64 {sr(S_SUB, 2)}")
        A("L_pc_%=:")
        for (reg, lab) in ((S_MASKFN[0], "L_mask0"), (S_MASKFN[1], "L_mask1"), (S_RESCFN[0], "L_resc0"),
                           (S_RESCFN[1], "L_resc1"), (S_ORESC, "L_oresc")):
            A(f"s_add_u32 s{reg}, s{S_SUB}, {lab}_%=-L_pc_%=")
            A(f"s_addc_u32 s{reg + 1}, s{S_SUB + 1}, 0")
        A(f"s_mov_b32 s{S_I}, s{S_IIN}")
        A(f"s_mov_b32 s{S_OOB}, 0x80000000")
        A(f"s_mov_b64 {sr(S_FLAG[0], 2)}, 0")
        A("s_barrier")                                     # previous pass is done with LDS
        if self.timers:
            for k in range(4):
                A(f"s_mov_b32 s{TM + k}, 0")
            L += self.stamp_lines(-1)
        A(f"s_cmp_eq_u32 s{S_ROLE}, 0")
        A("s_cbranch_scc0 L_orole_%=")

        # ================= S role =================
        if cfg.get("s_prio", 2):
            A(f"s_setprio {cfg.get('s_prio', 2)}")
        for qb in range(2):
            for ks in range(8):
                A(f"buffer_load_dwordx4 {ar(SA_Q[qb] + 4 * ks, 4)}, v{SV_QOFF[qb]}, {sr(S_QRS, 4)}, 0 offen offset:{32 * ks}")
        for qb in range(2):
            A(f"v_mov_b32 v{SV_MRUN[qb]}, 0xff800000")
            A(f"v_mov_b32 v{SV_NEGM[qb]}, 0x7f7fffff")
            A(f"v_mov_b32 v{SV_L[qb]}, 0")
            A(f"v_mov_b32 v{SV_L2[qb]}, 0")
            A(f"v_mov_b32 v{SV_THR[qb]}, 0xff800000")
        A("s_waitcnt vmcnt(0)")
        A("s_barrier")                                     # K(n_min) has landed (O waves)
        A("L_stop_%=:")
        A(f"s_cmp_ge_i32 s{S_I}, s{S_WLO}")
        A("s_cbranch_scc0 L_sidle_%=")
        A(f"s_cmp_lt_i32 s{S_I}, s{S_WHI}")
        A("s_cbranch_scc0 L_sidle_%=")
        A(f"s_bitcmp1_b32 s{S_I}, 0")
        A("s_cbranch_scc1 L_sact1_%=")
        report = {}
        for par in (0, 1):
            A(f"L_sact{par}_%=:")
            self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
            self.s_iteration(par, cfg)
            report[("S", par)] = (dict(self.stats), len(self.out))
            L += self.out
            A("s_branch L_stail_%=")
        A("L_sidle_%=:")
        A("L_stail_%=:")
        A("s_waitcnt lgkmcnt(0)")
        L += self.stamp_lines(3)
        A("s_barrier")
        A(f"s_add_u32 s{S_I}, s{S_I}, 1")
        A(f"s_cmp_le_i32 s{S_I}, s{S_NMAX}")
        A("s_cbranch_scc1 L_stop_%=")
        # S epilogue: LSE to memory, 1/l to the partner
        T = SV_T
        for qb in range(2):
            l, mrun = f"v{SV_L[qb]}", f"v{SV_MRUN[qb]}"
            ta, lt, inv, lse, zero = (f"v{T + i}" for i in range(5))
            A(f"v_add_f32 {l}, {l}, v{SV_L2[qb]}")
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
            A(f"buffer_store_dword {lse}, v{SV_LSEOFF[qb]}, {sr(S_LRS, 4)}, 0 offen")
            A(f"ds_write_b32 v{V_PALPHA}, {inv} offset:{256 * qb}")
        A("s_waitcnt lgkmcnt(0)")
        A("s_barrier")
        A("s_waitcnt vmcnt(0)")
        L += self.dump_timers(0, SV_LSEOFF[0])
        A("s_setprio 0")
        A("s_branch L_end_%=")

        # ================= O role =================
        A("L_orole_%=:")
        # K(n_min) -> its slot, by parity of n_min
        t = S_T
        A(f"s_mul_i32 s{t + 2}, s{S_I}, s{S_KTILE}")
        A(f"s_cmp_lt_i32 s{S_I}, s{S_NMAX}")
        A(f"s_cselect_b32 s{t + 2}, s{t + 2}, s{S_OOB}")
        A(f"s_bitcmp1_b32 s{S_I}, 0")
        A(f"s_cselect_b32 s{t}, {K_SLOT[1]}, {K_SLOT[0]}")
        A(f"s_add_u32 s{t}, s{t}, s{S_W1024}")
        for jj in range(4):
            A(f"s_add_u32 m0, s{t}, {4096 * jj}")
            A("s_nop 0")
            A(f"buffer_load_dwordx4 v{OV_DMAK}, {sr(S_KRS, 4)}, s{t + 2} offen lds")
            A(f"s_add_u32 s{t + 2}, s{t + 2}, s{S_K16}")
        for (p, b) in O_ACC:
            for i in range(16):
                A(f"v_accvgpr_write_b32 a{b + i}, 0" if p == "a" else f"v_mov_b32 v{b + i}, 0")
        A("s_waitcnt vmcnt(0)")
        A("s_barrier")
        A("L_otop_%=:")
        A(f"s_sub_u32 s{t}, s{S_I}, 1")
        A(f"s_cmp_ge_i32 s{t}, s{S_WLO}")
        A("s_cbranch_scc0 L_oidle_%=")
        A(f"s_cmp_lt_i32 s{t}, s{S_WHI}")
        A("s_cbranch_scc0 L_oidle_%=")
        A(f"s_bitcmp1_b32 s{S_I}, 0")
        A("s_cbranch_scc1 L_oact1_%=")
        for par in (0, 1):
            A(f"L_oact{par}_%=:")
            self.out, self.stats = [], {"nop_states": 0, "lgkm_waits": 0}
            self.o_iteration(par, True, cfg)
            report[("O", par)] = (dict(self.stats), len(self.out))
            L += self.out
            A("s_branch L_otail_%=")
        A("L_oidle_%=:")
        A(f"s_bitcmp1_b32 s{S_I}, 0")
        A("s_cbranch_scc1 L_oidle1_%=")
        for par in (0, 1):
            A(f"L_oidle{par}_%=:")
            self.out = []
            self.o_iteration(par, False, cfg)
            L += self.out
            if par == 0:
                A("s_branch L_otail_%=")
        A("L_otail_%=:")
        A("s_waitcnt vmcnt(0) lgkmcnt(0)")
        L += self.stamp_lines(3)
        A("s_barrier")
        A(f"s_add_u32 s{S_I}, s{S_I}, 1")
        A(f"s_cmp_le_i32 s{S_I}, s{S_NMAX}")
        A("s_cbranch_scc1 L_otop_%=")
        # O epilogue
        A("s_barrier")                                     # 1/l is in the P slot
        A("s_nop 7")
        A("s_nop 7")
        T = OV_T
        for qb in range(2):
            inv = f"v{T + qb}"
            A(f"ds_read_b32 {inv}, v{V_PALPHA} offset:{256 * qb}")
        A("s_waitcnt lgkmcnt(0)")
        for qb in range(2):
            inv = f"v{T + qb}"
            for d in range(4):
                p, b = O_ACC[qb * 4 + d]
                for r4 in range(4):
                    tt = T + 4 + 4 * (r4 & 1)
                    for e in range(4):
                        if p == "a":
                            A(f"v_accvgpr_read_b32 v{tt + e}, a{b + 4 * r4 + e}")
                        else:
                            A(f"v_mov_b32 v{tt + e}, v{b + 4 * r4 + e}")
                    for e in range(4):
                        A(f"v_mul_f32 v{tt + e}, v{tt + e}, {inv}")
                    pk = T + 12 + 2 * (r4 & 1)
                    A(f"{self.cvt} v{pk}, v{tt}, v{tt + 1}")
                    A(f"{self.cvt} v{pk + 1}, v{tt + 2}, v{tt + 3}")
                    A(f"buffer_store_dwordx2 {vr(pk, 2)}, v{OV_OOFF[qb]}, {sr(S_ORS, 4)}, 0 offen offset:{64 * d + 16 * r4}")
        A("s_waitcnt vmcnt(0)")
        L += self.dump_timers(8, 13)
        A("s_branch L_end_%=")

        # ================= routines =================
        for qb in range(2):
            A(f"L_mask{qb}_%=:")
            L += self.s_mask_routine(qb)
            A(f"L_resc{qb}_%=:")
            L += self.s_rescale_routine(qb)
        A("L_oresc_%=:")
        L += self.o_rescale_routine()
        A("L_end_%=:")
        return L, report


def clobbers():
    c = ["memory", "vcc", "scc", "m0"]
    c += [f"v{i}" for i in range(26, N_ARCH)]
    c += [f"a{i}" for i in range(N_ACC)]
    c += [f"s{i}" for i in range(S_I, S_LAST + 1)]
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
    print("// GENERATED by gen_fwd_ws.py - do not edit.  See that script for the design and the register map.")
    print("#pragma once")
    print(f"#define FA_FWD_WS_LDS_BYTES {LDS_TOTAL}")
    print(f"#define FA_FWD_WS_P_BASE {P_BASE}")
    print(f"#define FA_FWD_WS_P_PAIR_BYTES {2 * P_SLOT_BYTES}")
    print(f"#define FA_FWD_WS_P_ALPHA {P_ALPHA}")
    for dt in ("bf16", "f16"):
        g = WS(dt)
        g.ko = ko
        g.timers = bool(cfg.get("timers", 0))
        body, report = g.gen_body(cfg)
        print(f"#define FA_FWD_WS_BODY_{dt.upper()} \\")
        for ln in body:
            print(f'    "{ln}\\n" \\')
        print('    ""')
        for k, (st, n) in report.items():
            print(f"// {dt} {k}: {n} lines, nop states {st['nop_states']}, lgkmcnt waits {st['lgkm_waits']}")
    cl = ", ".join(f'"{c}"' for c in clobbers())
    print(f"#define FA_FWD_WS_CLOBBERS {cl}")


if __name__ == "__main__":
    main()
