#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of the relative-position flash-attention forward that keeps its probabilities
(bdm_db1_amd/csrc/relattn_flash_fwd2_loop.inc, used by relattn_flash_fwd2.hip; plain causal window only).

Why: the compiled loop (relattn_flash.hip) issues ~380 instructions per (16 queries x 32 keys) wave-block, 64 of them copies of the
output accumulators around its branches, drains the matrix pipe in front of every barrier and runs the two relative-term chains one
after the other: 2500 cycles per block where the 24 MFMAs need 384.  Here the block is ~165 instructions on physical registers, and
the score phase of block n+1 (relative-term tiles, S^T = K.Qu^T, skewed read) is software-pipelined against the softmax + P.V phase
of block n, so that every MFMA has VALU / LDS work of the OTHER phase to issue behind it (counted lgkmcnt waits, no drain).

Pipeline (r = iteration, one s_barrier per iteration; K tile of block n in stage n % 3, V likewise, ring rows by distance):
  iteration r:   A(r)  = relative-term tiles + S^T + skewed read of block r            -> s_next (8 raw scores per lane), block maximum
                 B(r-1) = exp2 / row sums / p~ pack + store / O^T += V^T.P^T of block r-1 (its scores: s_cur, maximum known)
  prefetch at iteration r: K[r+2] -> stage (r+2) % 3, V[r+1] -> stage (r+1) % 3, the 32 ring rows block r+2 adds  (always issued:
  beyond the last block the pointers stop advancing and the pieces land in stages nobody reads).
A wave (16 queries from iw) sees blocks 0 .. last = iw / 32; block `last` is its diagonal block (element mask).  Per iteration it takes
ONE of four paths (wave-uniform): r == 0: A only (all three band tiles); 1 <= r <= last: A(r) | B(r-1) interleaved (the steady
state); r == last + 1: B(last) only; later: idle (its LDS-DMA pieces and the barrier).  The code is unrolled 6 x (stage = r % 3 and
band parity = r % 2 are immediates).

The script keeps the LDS queue (in-order per wave) and inserts the counted `s_waitcnt lgkmcnt(n)` itself; it also checks / pads the
software-visible hazards of gfx950 (MFMA result -> VALU / LDS / VMEM use, VALU -> MFMA / permlane, trans -> use).
"""
import os
import sys

# ---------------------------------------------------------------------------------------------- register map (VGPR numbers)
O = 0          # acc_o[8][4]          O^T accumulators
QU = 32        # fqu[4][4]
QV = 48        # fqv[4][4]
F = 64         # fragment bank: 8 x 4 (ring fragments, then K fragments)
VT = 96        # V^T fragments 8 x 4 (A-first: the third ring tile's fragments in VT..VT+15)
T = 128        # relative-term accumulators, tiles 0 / 1
AS = 136       # S^T accumulators, tiles 0 / 1
SK = 144       # skewed relative term (8)   (A-first: SK..SK+3 = accumulator of the third band tile)
S0 = 152       # score set 0 (8)
S1 = 160       # score set 1 (8)
PB = 168       # p~ as bf16x8 (4)
ROWF = 172     # [t][ks] (8)
TR = 180       # [t][db] (16)
RING = 196     # [ks] (4)
TSK = 200      # [par][8] (16)
TWR = 216
SROW = 217     # wave * 4 + (lane >> 4)
THR = 218      # thr[t] (2): mask thresholds of the diagonal block
NEG = 220      # -1e30
KP = 222       # K source pointer (2)
VP = 224
RB = 226       # ring source base (2)
PP = 228       # p~ image pointer (2)
MP = 230       # block-maximum pointer (2)
MI = 232
LI = 233
MBLK = 234     # maximum of the block whose scores are in s_cur
MC = 235
T0 = 236       # temporaries T0..T0+9
RA = 246       # ring read addresses (8): 246..253
NV = 254

# SGPRs
S_M0 = 40
S_R = 41
S_END = 42
S_LAST = 43
S_LDSK = 44
S_LDSV = 45
S_LDSR = 46
S_RDIST = 47     # distance of the first ring row of the next request (lane part: + srow)
S_RDW8 = 48      # (that distance + 4 wave) << 8
S_DLO8 = 49      # (iw - 32 r - 32) << 8: band start of block r
S_RS0 = 50
S_RS1 = 51
S_KST = 52       # pair: current K / V pointer step (0 once the last tile has been requested)
S_VST = 54
S_STEP = 56      # pair: one key block in bytes
S_RRS = 58
S_LM1 = 59
S_PST = 60       # pair
S_MST = 62       # pair
S_C2 = 64
S_T0 = 65
S_T1 = 66
S_JH = 67        # jb_hi
S_X = 68         # pair scratch (carry-out of v_mad_i64_i32)
S_RS2 = 70

OFF_K = 0
OFF_V = 3 * 8192
OFF_R = 6 * 8192
TP = 68


class I:
    __slots__ = ("text", "kind", "rd", "wr", "rare", "srcc")

    def __init__(self, text, kind, rd=(), wr=(), rare=False, srcc=()):
        self.text, self.kind, self.rd, self.wr, self.rare, self.srcc = text, kind, frozenset(rd), frozenset(wr), rare, frozenset(srcc)


def rng(b, n):
    return range(b, b + n)


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def sr(b, n=1):
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


# ---------------------------------------------------------------------------------------------- instruction constructors
def mfma(dst, a, b, c=None):
    cs = "0" if c is None else vr(c, 4)
    rd = set(rng(a, 4)) | set(rng(b, 4)) | (set(rng(c, 4)) if c is not None else set())
    return I(f"v_mfma_f32_16x16x32_bf16 {vr(dst, 4)}, {vr(a, 4)}, {vr(b, 4)}, {cs}", "mfma", rd, rng(dst, 4), srcc=(rng(c, 4) if c is not None else ()))


def valu(op, dst, *src, kind="valu", extra_rd=()):
    """src: ints = VGPRs, strings = literals / SGPRs"""
    ops = ", ".join(vr(x) if isinstance(x, int) else x for x in src)
    rd = {x for x in src if isinstance(x, int)} | set(extra_rd)
    return I(f"{op} {vr(dst)}, {ops}", kind, rd, {dst})


def ds_read_b128(dst, addr, off):
    return I(f"ds_read_b128 {vr(dst, 4)}, {vr(addr)} offset:{off}", "dsr", {addr}, rng(dst, 4))


def ds_read_tr(dst, addr, off):
    return I(f"ds_read_b64_tr_b16 {vr(dst, 2)}, {vr(addr)} offset:{off}", "dsr", {addr}, rng(dst, 2))


def ds_read_b32(dst, addr, off=0):
    return I(f"ds_read_b32 {vr(dst)}, {vr(addr)}" + (f" offset:{off}" if off else ""), "dsr", {addr}, {dst})


def ds_write2(addr, d0, d1, o0, o1):
    assert 0 <= o0 < 256 and 0 <= o1 < 256
    return I(f"ds_write2_b32 {vr(addr)}, {vr(d0)}, {vr(d1)} offset0:{o0} offset1:{o1}", "dsw", {addr, d0, d1}, ())


def salu(text):
    return I(text, "salu")


def nop(n):
    return I(f"s_nop {n}", "nop")


def states(ins):
    if ins.kind == "nop":
        return int(ins.text.split()[1]) + 1
    if ins.kind in ("label", "wait"):
        return 0 if ins.kind == "label" else 1
    return 1


# ---------------------------------------------------------------------------------------------- hazards + counted lgkmcnt
def need_states(w, rdr, reg_is_srcc):
    """minimum wait states between writer w and a later instruction rdr touching one of w's destination registers"""
    if w.kind == "mfma":
        if rdr.kind == "mfma" and reg_is_srcc:
            return 0          # accumulate chain: hardware-interlocked
        return 10             # XDL write -> VALU / LDS / VMEM / MFMA A,B (hipcc pads 8 for this opcode; two spare)
    if w.kind in ("valu", "trans", "perm"):
        if rdr.kind == "mfma":
            return 2
        if rdr.kind == "perm":
            return 2
        if w.kind == "trans":
            return 2
    return 0


def finalize(seq, name):
    """insert s_waitcnt lgkmcnt(n) / s_nop where needed; seq: list of I (straight line; `rare` instructions may or may not execute,
    they contain no LDS operations).  Returns the new list."""
    out = []
    queue = []          # outstanding LDS operations of this path: sets of destination registers (empty for writes), oldest first
    pads = 0

    def dist_back(idx_writer):
        # wait states between out[idx_writer] and the instruction about to be appended, counting rare instructions as absent
        return sum(states(x) for x in out[idx_writer + 1:] if not x.rare)

    last_writer = {}    # vgpr -> index in out
    look = int(os.environ.get("FW_WAIT_LOOKAHEAD", 8))
    for idx, ins in enumerate(seq):
        touched = ins.rd | ins.wr
        # LDS results
        need = None
        for pos, dst in enumerate(queue):
            if dst & touched:
                need = pos
        if need is not None:
            # one wait for a run of consumers: also cover what the next few instructions need, as long as no LDS / memory operation or
            # branch comes first (every s_waitcnt is an issue slot of an issue-bound wave)
            for nx in seq[idx + 1: idx + 1 + look]:
                if nx.kind in ("dsr", "dsw", "vmem", "branch", "label", "barrier", "wait"):
                    break
                t2 = nx.rd | nx.wr
                for pos, dst in enumerate(queue):
                    if dst & t2 and pos > need:
                        need = pos
            after = min(len(queue) - 1 - need, 15)    # (a smaller count only waits longer: the queue retires in order)
            out.append(I(f"s_waitcnt lgkmcnt({after})", "wait"))
            queue = queue[len(queue) - after:] if after else []
        # software hazards
        worst = 0
        for r in touched:
            if r in last_writer:
                w = out[last_writer[r]]
                req = need_states(w, ins, ins.kind == "mfma" and w.kind == "mfma" and r in ins.srcc and r in ins.wr)   # (an accumulate chain reads its own destination as SrcC)
                if req:
                    have = dist_back(last_writer[r])
                    worst = max(worst, req - have)
        if worst > 0:
            pads += worst
            while worst > 0:
                k = min(worst, 16)
                out.append(nop(k - 1))
                worst -= k
        out.append(ins)
        if ins.kind in ("dsr", "dsw"):
            queue.append(frozenset(ins.wr))
            assert len(queue) <= 40
        for r in ins.wr:
            last_writer[r] = len(out) - 1
    return out, pads


# ---------------------------------------------------------------------------------------------- building blocks
def dma(r6):
    """the three LDS-DMA requests of iteration r (r % 6 == r6) + pointer steps; returns instruction groups (each group stays together)"""
    kst, vst = (r6 + 2) % 3, (r6 + 1) % 3
    g = []
    g.append([salu(f"s_add_u32 m0, {sr(S_LDSK)}, {kst * 8192}"), nop(0),
              I(f"global_load_lds_dwordx4 {vr(KP, 2)}, off", "vmem", {KP, KP + 1}, ()),
              I(f"v_lshl_add_u64 {vr(KP, 2)}, {vr(KP, 2)}, 0, {sr(S_KST, 2)}", "valu", {KP, KP + 1}, {KP, KP + 1})])
    g.append([salu(f"s_add_u32 m0, {sr(S_LDSV)}, {vst * 8192}"), nop(0),
              I(f"global_load_lds_dwordx4 {vr(VP, 2)}, off", "vmem", {VP, VP + 1}, ()),
              I(f"v_lshl_add_u64 {vr(VP, 2)}, {vr(VP, 2)}, 0, {sr(S_VST, 2)}", "valu", {VP, VP + 1}, {VP, VP + 1})])
    g.append([valu("v_add_u32", T0 + 8, sr(S_RDIST), SROW),
              I(f"v_med3_i32 {vr(T0 + 8)}, {vr(T0 + 8)}, 0, {sr(S_LM1)}", "valu", {T0 + 8}, {T0 + 8}),
              I(f"v_mad_i64_i32 {vr(T0 + 8, 2)}, {sr(S_X, 2)}, {vr(T0 + 8)}, {sr(S_RRS)}, {vr(RB, 2)}", "valu", {T0 + 8, RB, RB + 1}, {T0 + 8, T0 + 9}),
              salu(f"s_and_b32 {sr(S_T0)}, {sr(S_RDW8)}, 0xff00"),
              salu(f"s_add_u32 m0, {sr(S_T0)}, {sr(S_LDSR)}"), nop(0),
              I(f"global_load_lds_dwordx4 {vr(T0 + 8, 2)}, off", "vmem", {T0 + 8, T0 + 9}, ()),
              salu(f"s_sub_u32 {sr(S_RDIST)}, {sr(S_RDIST)}, 32"),
              salu(f"s_sub_u32 {sr(S_RDW8)}, {sr(S_RDW8)}, 0x2000")])
    return g


def ptr_clamps():
    """K step -> 0 once block r + 3 > jb_hi, V step -> 0 once r + 2 > jb_hi (evaluated before this iteration's requests advance the pointers)"""
    return [salu(f"s_add_u32 {sr(S_T0)}, {sr(S_R)}, 3"),
            salu(f"s_cmp_le_u32 {sr(S_T0)}, {sr(S_JH)}"),
            salu(f"s_cselect_b64 {sr(S_KST, 2)}, {sr(S_STEP, 2)}, 0"),
            salu(f"s_add_u32 {sr(S_T0)}, {sr(S_R)}, 2"),
            salu(f"s_cmp_le_u32 {sr(S_T0)}, {sr(S_JH)}"),
            salu(f"s_cselect_b64 {sr(S_VST, 2)}, {sr(S_STEP, 2)}, 0")]


def band_scalars(ntiles):
    """ring offsets of the band tiles of block r from S_DLO8 = (iw - 32 r - 32) << 8, then step it to the next block"""
    o = [salu(f"s_and_b32 {sr(S_RS0)}, {sr(S_DLO8)}, 0xff00"),
         salu(f"s_add_u32 {sr(S_T0)}, {sr(S_DLO8)}, 0x1000"),
         salu(f"s_and_b32 {sr(S_RS1)}, {sr(S_T0)}, 0xff00")]
    if ntiles == 3:
        o += [salu(f"s_add_u32 {sr(S_T0)}, {sr(S_DLO8)}, 0x2000"), salu(f"s_and_b32 {sr(S_RS2)}, {sr(S_T0)}, 0xff00")]
    o.append(salu(f"s_sub_u32 {sr(S_DLO8)}, {sr(S_DLO8)}, 0x2000"))
    return o


def ring_addr(tt, ks, dst):
    return valu("v_add_u32", dst, sr((S_RS0, S_RS1, S_RS2)[tt]), RING + ks)


def softmax_top(cur):
    """running maximum update + (rare) rescale of O and l, then mc = -m c2"""
    # deferred maximum: the running maximum only moves when a block exceeds it by more than 2^THR (in exp2 units); p~ = exp2((s - m) c2)
    # is then bounded by 2^THR instead of 1, which fp32 sums and bf16 p~ (a relative format) do not mind, and the stored block maximum is
    # the one that was USED.  With the exact rule the O-wide rescale ran in most iterations (some row of 16 finds a new maximum).
    thr = float(os.environ.get("FW2_THR", 8.0))
    import struct
    thr_hex = "0x%08x" % struct.unpack("<I", struct.pack("<f", thr))[0]
    o = [valu("v_sub_f32", T0, MBLK, MI),
         valu("v_mul_f32", T0, sr(S_C2), T0),
         I(f"v_cmp_lt_f32 vcc, {thr_hex}, {vr(T0)}", "valu", {T0}, ()),
         salu("s_cmp_eq_u64 vcc, 0"),
         I("s_cbranch_scc1 L_nr_@", "branch")]
    rare = [valu("v_max_f32", T0, MI, MBLK), valu("v_sub_f32", T0 + 1, MI, T0), valu("v_mul_f32", T0 + 1, sr(S_C2), T0 + 1), valu("v_exp_f32", T0 + 1, T0 + 1, kind="trans"),
            valu("v_mov_b32", MI, T0), nop(1), valu("v_mul_f32", LI, LI, T0 + 1)]
    rare += [valu("v_mul_f32", O + i, O + i, T0 + 1) for i in range(32)]
    for x in rare:
        x.rare = True
    o += rare
    o.append(I("L_nr_@:", "label"))
    o.append(I(f"v_mul_f32_e64 {vr(MC)}, -{vr(MI)}, {sr(S_C2)}", "valu", {MI}, {MC}))
    return o


def exp_pairs(cur):
    return [[I(f"v_fma_f32 {vr(cur + i)}, {vr(cur + i)}, {sr(S_C2)}, {vr(MC)}", "valu", {cur + i, MC}, {cur + i}),
             valu("v_exp_f32", cur + i, cur + i, kind="trans")] for i in range(8)]


def lsum_cvt(cur):
    o = [valu("v_add_f32", T0 + 2, cur + 0, cur + 1), valu("v_add_f32", T0 + 3, cur + 2, cur + 3),
         valu("v_add_f32", T0 + 4, cur + 4, cur + 5), valu("v_add_f32", T0 + 5, cur + 6, cur + 7),
         valu("v_cvt_pk_bf16_f32", PB + 0, cur + 0, cur + 1), valu("v_cvt_pk_bf16_f32", PB + 1, cur + 2, cur + 3),
         valu("v_add_f32", T0 + 2, T0 + 2, T0 + 3), valu("v_add_f32", T0 + 4, T0 + 4, T0 + 5),
         valu("v_cvt_pk_bf16_f32", PB + 2, cur + 4, cur + 5), valu("v_cvt_pk_bf16_f32", PB + 3, cur + 6, cur + 7),
         valu("v_add_f32", T0 + 2, T0 + 2, T0 + 4), valu("v_add_f32", LI, LI, T0 + 2)]
    return o


def stores():
    return [I(f"global_store_dwordx4 {vr(PP, 2)}, {vr(PB, 4)}, off", "vmem", set(rng(PB, 4)) | {PP, PP + 1}, ()),
            I(f"v_mul_f32 {vr(T0 + 6)}, {sr(S_C2)}, {vr(MI)}", "valu", {MI}, {T0 + 6}),
            I(f"v_lshl_add_u64 {vr(PP, 2)}, {vr(PP, 2)}, 0, {sr(S_PST, 2)}", "valu", {PP, PP + 1}, {PP, PP + 1}),
            salu(f"s_sub_u32 {sr(S_PST)}, {sr(S_PST)}, 2048"),       # the p~ images are a triangle (relattn_flash.h): the row of the next key block is two tiles shorter
            salu("s_mov_b64 exec, 0xffff"),
            I(f"global_store_dword {vr(MP, 2)}, {vr(T0 + 6)}, off", "vmem", {MP, MP + 1, T0 + 6}, ()),
            salu("s_mov_b64 exec, -1"),
            I(f"v_lshl_add_u64 {vr(MP, 2)}, {vr(MP, 2)}, 0, {sr(S_MST, 2)}", "valu", {MP, MP + 1}, {MP, MP + 1})]


def vt_reads(stg):
    o = []
    for db in range(8):
        o.append(ds_read_tr(VT + 4 * db, TR + db, OFF_V + stg * 8192))
        o.append(ds_read_tr(VT + 4 * db + 2, TR + 8 + db, OFF_V + stg * 8192))
    return o


def k_read(t, ks, stg):
    return ds_read_b128(F + 4 * (4 * t + ks), ROWF + 4 * t + ks, OFF_K + stg * 8192)


def scratch_writes(acc, col):
    return [ds_write2(TWR, acc + 0, acc + 1, col, col + TP), ds_write2(TWR, acc + 2, acc + 3, col + 2 * TP, col + 3 * TP)]


def skew_reads(par):
    return [ds_read_b32(SK + i, TSK + 8 * par + i) for i in range(8)]


def next_scores(nxt):
    return [valu("v_add_f32", nxt + i, AS + i, SK + i) for i in range(8)]


def mask_block(nxt):
    """diagonal block: key kk(t, g) + r visible iff r <= thr[t]"""
    o = [salu(f"s_cmp_lg_u32 {sr(S_R)}, {sr(S_LAST)}"), I("s_cbranch_scc1 L_nm_@", "branch")]
    rare = []
    for t in range(2):
        for r in range(4):
            rare.append(I(f"v_cmp_le_i32 vcc, {r}, {vr(THR + t)}", "valu", {THR + t}, ()))
            rare.append(I(f"v_cndmask_b32 {vr(nxt + 4 * t + r)}, {vr(NEG)}, {vr(nxt + 4 * t + r)}, vcc", "valu", {NEG, nxt + 4 * t + r}, {nxt + 4 * t + r}))
    for x in rare:
        x.rare = True
    return o + rare + [I("L_nm_@:", "label")]


def block_max(nxt):
    o = [I(f"v_max3_f32 {vr(T0)}, {vr(nxt)}, {vr(nxt + 1)}, {vr(nxt + 2)}", "valu", {nxt, nxt + 1, nxt + 2}, {T0}),
         I(f"v_max3_f32 {vr(T0 + 1)}, {vr(nxt + 3)}, {vr(nxt + 4)}, {vr(nxt + 5)}", "valu", {nxt + 3, nxt + 4, nxt + 5}, {T0 + 1}),
         valu("v_max_f32", T0 + 2, nxt + 6, nxt + 7),
         I(f"v_max3_f32 {vr(MBLK)}, {vr(T0)}, {vr(T0 + 1)}, {vr(T0 + 2)}", "valu", {T0, T0 + 1, T0 + 2}, {MBLK})]
    for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
        o += [valu("v_mov_b32", T0, MBLK),
              I(f"{op} {vr(MBLK)}, {vr(T0)}", "perm", {MBLK, T0}, {MBLK, T0}),
              valu("v_max_f32", MBLK, MBLK, T0)]
    return o


def interleave(slots, fillers, per=None):
    """slots: list of MFMA instructions; fillers: list of groups (lists) spread evenly behind the slots (group i behind slot
    floor(i * len(slots) / len(groups)) unless `per` gives the slot of each group)"""
    out = []
    n, m = len(slots), len(fillers)
    where = per if per is not None else [min(n - 1, (i * n) // max(m, 1)) for i in range(m)]
    for s, ins in enumerate(slots):
        out.append(ins)
        for gi, grp in enumerate(fillers):
            if where[gi] == s:
                out += grp
    return out


def tail(nstores, r6, prev_stores=0):
    # everything but this iteration's requests / stores (and the previous iteration's stores, which are younger than ITS requests) has to
    # have landed: the queue retires in order, so the stores get one more iteration for their acknowledgements
    return [I(f"s_waitcnt vmcnt({3 + nstores + prev_stores})", "wait"), I("s_barrier", "barrier"),
            salu(f"s_add_u32 {sr(S_R)}, {sr(S_R)}, 1"),
            salu(f"s_cmp_gt_u32 {sr(S_R)}, {sr(S_END)}"),
            I("s_cbranch_scc1 L_done_%=", "branch"),
            I(f"s_branch L_inst{(r6 + 1) % 6}_%=", "branch")]


# ---------------------------------------------------------------------------------------------- the four paths of one instance
def path_first(r6):
    """r == 0: band tiles 0, 1, 2 + S^T + skewed read of block 0 (stage 0, parity 0); no softmax / P.V phase"""
    assert r6 == 0
    seq = band_scalars(3) + ptr_clamps()
    fb = [F + 4 * (4 * tt + ks) if tt < 2 else VT + 4 * ks for tt in range(3) for ks in range(4)]    # fragment registers [tt][ks]
    acc = [T, T + 4, SK]
    for tt in range(3):
        for ks in range(4):
            seq.append(ring_addr(tt, ks, RA + (4 * tt + ks) % 8))
            seq.append(ds_read_b128(fb[4 * tt + ks], RA + (4 * tt + ks) % 8, OFF_R))
    rel = [mfma(acc[tt], QV + 4 * ks, fb[4 * tt + ks], None if ks == 0 else acc[tt]) for ks in range(4) for tt in range(3)]
    d = dma(r6)
    fill = [d[0], d[1], d[2]] + [[k_read(t, ks, 0)] for ks in range(4) for t in range(2)]
    per = [0, 2, 4] + [3 + i for i in range(8)]       # K fragment (t, ks) overwrites the ring fragment of tile t, ks: its MFMA is slot 3 ks + t <= 3 + i
    seq += interleave(rel, fill, per)
    for tt in range(3):
        seq += scratch_writes(acc[tt], 16 * tt)
    smf = [mfma(AS + 4 * t, F + 4 * (4 * t + ks), QU + 4 * ks, None if ks == 0 else AS + 4 * t) for ks in range(4) for t in range(2)]
    seq += smf
    seq += skew_reads(0)
    seq += next_scores(S0) + mask_block(S0) + block_max(S0)
    seq += tail(0, r6)
    return seq


def path_steady(r6):
    stg_a, par, stg_b = r6 % 3, r6 % 2, (r6 + 2) % 3
    cur, nxt = (S1, S0) if par == 0 else (S0, S1)       # block r's scores go to set par; block r-1's are in the other set
    seq = band_scalars(2) + ptr_clamps()
    for tt in range(2):
        for ks in range(4):
            seq.append(ring_addr(tt, ks, RA + 4 * tt + ks))
    for tt in range(2):
        for ks in range(4):
            seq.append(ds_read_b128(F + 4 * (4 * tt + ks), RA + 4 * tt + ks, OFF_R))
    seq += softmax_top(cur)
    # relative-term MFMAs; behind them: K fragment reads into the freed slots, the LDS-DMA requests, exp2 of block r-1
    rel = [mfma(T + 4 * tt, QV + 4 * ks, F + 4 * (4 * tt + ks), None if ks == 0 else T + 4 * tt) for ks in range(4) for tt in range(2)]
    ex = exp_pairs(cur)
    d = dma(r6)
    fill, per = [], []
    for i in range(8):
        ks, t = i >> 1, i & 1
        fill.append([k_read(t, ks, stg_a)] + ex[i])
        per.append(i)
    for j, grp in enumerate(d):
        fill.append(grp)
        per.append(1 + 2 * j)
    seq += interleave(rel, fill, per)
    # S^T MFMAs; behind them: row sums + pack, scratch writes, skewed reads, V^T fragment reads, the two stores
    smf = [mfma(AS + 4 * t, F + 4 * (4 * t + ks), QU + 4 * ks, None if ks == 0 else AS + 4 * t) for ks in range(4) for t in range(2)]
    lc = lsum_cvt(cur)
    vtr = vt_reads(stg_b)
    sw = scratch_writes(T, (0) ^ (32 * par)) + scratch_writes(T + 4, (16) ^ (32 * par))
    skr = skew_reads(par)
    st = stores()
    fill = [lc[0:6] + vtr[0:2], lc[6:12] + vtr[2:4], sw[0:2] + vtr[4:6], sw[2:4] + vtr[6:8],
            skr[0:4] + vtr[8:10], skr[4:8] + vtr[10:12], st[0:4] + vtr[12:14], st[4:8] + vtr[14:16]]
    seq += interleave(smf, fill, list(range(8)))
    # P.V MFMAs; behind them: the next block's scores, its mask (diagonal block only) and its maximum
    pv = [mfma(O + 4 * db, VT + 4 * db, PB, O + 4 * db) for db in range(8)]
    bm = block_max(nxt)
    fill = [next_scores(nxt), mask_block(nxt), bm[0:4], bm[4:7], bm[7:10]]     # (the mask block is skipped by a branch: it stays in one piece)
    seq += interleave(pv, fill, [3, 4, 5, 6, 7])
    seq += tail(2, r6, int(os.environ.get('FW2_PREVST', 2)) if r6 != 1 else 0)
    return seq


def path_last(r6):
    """r == last + 1: softmax + P.V of the wave's diagonal block (already masked), no score phase"""
    par, stg_b = r6 % 2, (r6 + 2) % 3
    cur = S1 if par == 0 else S0
    seq = ptr_clamps() + [x for g in dma(r6) for x in g]
    seq += vt_reads(stg_b)
    seq += softmax_top(cur)
    for p in exp_pairs(cur):
        seq += p
    seq += lsum_cvt(cur) + stores()
    seq += [mfma(O + 4 * db, VT + 4 * db, PB, O + 4 * db) for db in range(8)]
    seq += tail(2, r6)
    return seq


def path_idle(r6):
    return ptr_clamps() + [x for g in dma(r6) for x in g] + tail(0, r6)


def instance(r6):
    out = [I(f"L_inst{r6}_%=:", "label")]
    total_pads = 0

    def add(seq, tag):
        nonlocal total_pads
        fin, pads = finalize(seq, tag)
        total_pads += pads
        for x in fin:
            x.text = x.text.replace("_@", f"_{tag}_%=")
        out.extend(fin)

    if r6 == 0:
        out += [salu(f"s_cmp_lg_u32 {sr(S_R)}, 0"), I("s_cbranch_scc1 L_nf_%=", "branch")]
        add(path_first(0), "f0")
        out.append(I("L_nf_%=:", "label"))
    out += [salu(f"s_cmp_le_u32 {sr(S_R)}, {sr(S_LAST)}"), I(f"s_cbranch_scc0 L_ns{r6}_%=", "branch")]
    add(path_steady(r6), f"s{r6}")
    out += [I(f"L_ns{r6}_%=:", "label"), salu(f"s_add_u32 {sr(S_T1)}, {sr(S_LAST)}, 1"), salu(f"s_cmp_lg_u32 {sr(S_R)}, {sr(S_T1)}"),
            I(f"s_cbranch_scc1 L_nl{r6}_%=", "branch")]
    add(path_last(r6), f"l{r6}")
    out.append(I(f"L_nl{r6}_%=:", "label"))
    add(path_idle(r6), f"i{r6}")
    return out, total_pads


def prologue():
    o = [salu(f"s_mov_b32 {sr(S_M0)}, m0"),
         salu(f"s_mov_b32 {sr(S_R)}, 0"),
         salu(f"s_mov_b32 {sr(S_JH)}, %[jbhi]"),
         salu(f"s_add_u32 {sr(S_END)}, %[jbhi], 1"),
         salu(f"s_mov_b32 {sr(S_LAST)}, %[last]"),
         salu(f"s_mov_b32 {sr(S_LDSK)}, %[ldsk]"),
         salu(f"s_add_u32 {sr(S_LDSV)}, %[ldsk], {OFF_V}"),
         salu(f"s_mov_b32 {sr(S_LDSR)}, %[ldsr]"),
         salu(f"s_mov_b32 {sr(S_RDIST)}, %[rdist]"),
         salu(f"s_mov_b32 {sr(S_RDW8)}, %[rdw8]"),
         salu(f"s_mov_b32 {sr(S_DLO8)}, %[dlo8]"),
         salu(f"s_mov_b64 {sr(S_STEP, 2)}, %[kvstep]"),
         salu(f"s_mov_b32 {sr(S_RRS)}, %[rrs]"),
         salu(f"s_mov_b32 {sr(S_LM1)}, %[lm1]"),
         salu(f"s_mov_b64 {sr(S_PST, 2)}, %[ptstep]"),
         salu(f"s_add_u32 {sr(S_T0)}, %[lm1], 1"),
         salu(f"s_lshl_b32 {sr(S_MST)}, {sr(S_T0)}, 2"),
         salu(f"s_mov_b32 {sr(S_MST + 1)}, 0"),
         salu(f"v_readfirstlane_b32 {sr(S_C2)}, %[c2]")]
    o += [I(f"v_mov_b32 {vr(O + i)}, 0", "valu", (), {O + i}) for i in range(32)]
    o += [I(f"v_mov_b32 {vr(MI)}, 0xf149f2ca", "valu", (), {MI}),      # -1e30f
          I(f"v_mov_b32 {vr(NEG)}, 0xf149f2ca", "valu", (), {NEG}),
          I(f"v_mov_b32 {vr(LI)}, 0", "valu", (), {LI}),
          I(f"v_mov_b32 {vr(MBLK)}, 0xf149f2ca", "valu", (), {MBLK})]
    return o


def epilogue():
    return [I("L_done_%=:", "label"), I("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait"), nop(15), salu(f"s_mov_b32 m0, {sr(S_M0)}")]


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bdm_db1_amd", "csrc")
    lines = prologue()
    pads = 0
    for r6 in range(6):
        ins, p = instance(r6)
        lines += ins
        pads += p
    lines += epilogue()
    abl = os.environ.get("FW2_ABLATE", "")   # timing-only ablations (results wrong): dma, store, mfma, lds, valu, barrier (comma-separated)
    if abl:
        keep_wr = set(range(KP, KP + 2)) | set(range(VP, VP + 2)) | set(range(PP, PP + 2)) | set(range(MP, MP + 2)) | {T0 + 8, T0 + 9} | set(range(RA, RA + 8))

        def keep(x):
            if "dma" in abl and x.text.startswith("global_load_lds"):
                return False
            if "store" in abl and x.text.startswith("global_store"):
                return False
            if "mfma" in abl and x.kind == "mfma":
                return False
            if "lds" in abl and x.kind in ("dsr", "dsw"):
                return False
            if "barrier" in abl and x.kind == "barrier":
                return False
            if "valu" in abl and x.kind in ("valu", "trans", "perm") and not (set(x.wr) & keep_wr) and not x.text.startswith("v_mov_b32 v" ) :
                return False
            return True
        lines = [x for x in lines if keep(x)]
    path = os.path.join(d, "relattn_flash_fwd2_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_flash_fwd.py -- do not edit; the schedule and the register map are described there.\n")
        for x in lines:
            f.write(f'"{x.text}\\n\\t"\n')
    n_steady = len(finalize(path_steady(1), "x")[0])
    print(f"{path}: {len(lines)} instructions, steady path {n_steady}, hazard pads {pads} states")


if __name__ == "__main__":
    main()
