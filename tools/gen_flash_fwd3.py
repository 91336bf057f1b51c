#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of the relative-position flash-attention forward that keeps its probabilities,
FOUR waves x 32 query rows (two 16-row tiles per wave, one wave per SIMD, the whole 512-entry register file):
bdm_db1_amd/csrc/relattn_flash_fwd3_loop.inc, used by relattn_flash_fwd3.hip (plain causal window only).

Why this shape (measured on the 8-wave version of this loop, tools/gen_flash_fwd.py): two waves per SIMD running the same
stream between the same barriers do not overlap -- with waves 4-7 switched off the kernel took 587 us instead of 997 us for half the rows,
each wave issuing 47 % of its cycles and waiting (LDS / MFMA results, barriers) the rest.  The independent work that fills those
waits has to come from the SAME wave: here a wave owns two query tiles, so every chain (relative-term band tiles, S^T, softmax, P.V)
exists twice and the two copies are interleaved; K / V^T fragments are read once for both tiles, three relative-position fragment
sets serve four (tile, distance-tile) products, and the per-iteration bookkeeping (LDS-DMA requests, scalars, barrier) is paid once per
32 rows.  Fragments, Q operands and the O accumulators live in AGPRs (MFMA / LDS operands only), the softmax in VGPRs.

Pipeline, paths, LDS layout, p~ / block-maximum images: as tools/gen_flash_fwd.py (iteration r: score phase A(r) of block r | softmax +
P.V phase B(r-1) of block r-1; requests K[r+2], V[r+1], ring rows of block r+2; r == 0 / 1 <= r <= last / r == last+1 / later).
A wave's two tiles share their diagonal block `last` (its first query row is a multiple of 32).
The row sums come out of the matrix pipe (L += ones . P^T, one more MFMA per tile instead of 8 VALU adds and two lane swaps), the
running maximum is deferred (tools/gen_flash_fwd.py, softmax_top).
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_flash_fwd import I, rng, sr, salu, nop, finalize, interleave  # noqa: E402

A0 = 1000      # register ids >= A0 are AGPRs


def R(b, n=1):
    p, b = ("a", b - A0) if b >= A0 else ("v", b)
    return f"{p}{b}" if n == 1 else f"{p}[{b}:{b + n - 1}]"


# ---------------------------------------------------------------------------------------------- register map
def AO(x, db=0):
    return A0 + 36 * x + 4 * db      # O^T accumulators of tile x


def AL(x):
    return A0 + 36 * x + 32          # row-sum accumulator of tile x


def AQU(x, ks):
    return A0 + 72 + 16 * x + 4 * ks


def AQV(x, ks):
    return A0 + 104 + 16 * x + 4 * ks


def AK(t, ks):
    return A0 + 136 + 4 * (4 * t + ks)


def AR(d, ks, r6=0):                 # relative-position fragments of distance tile d (0..2; 3: the V^T bank, first iteration only)
    # The three fragment sets ROTATE with the iteration: block r + 1 needs the distance tiles of block r moved down by 32, so its highest
    # tile (d = 2) IS block r's lowest (d = 0) -- physical set (d + r) mod 3 makes that the same registers, and the steady path reads
    # only two tiles from the ring (8 ds_read_b128 instead of 12 per iteration: a ninth of the loop's LDS bytes).
    return A0 + 168 + 4 * (4 * ((d + r6) % 3) + ks) if d < 3 else AVT(ks)


def AVT(db):
    return A0 + 216 + 4 * db


AONES = A0 + 248

T = 0            # relative-term accumulators: chain c -> T + 4 c;  c = 0: (tile 0, D0)  1: (0, D1)  2: (1, D1)  3: (1, D2)


def AS(x, t=0):
    return 16 + 4 * (2 * x + t)


def SK(x):
    return 32 + 8 * x


def S(x, st):
    return 48 + 16 * x + 8 * st


def PB(x):
    return 80 + 4 * x


ROWF, TR, RING, TSK = 88, 96, 112, 116
TWR = 132        # + x
SROW = 134
THR = 135        # + t
NEG = 137
KOFF = 138       # + p: lane byte offset of DMA piece p inside a K / V tile
ROFF = 140       # + p: lane byte offset of its chunk inside a ring row
PP = 142         # pair: p~ image pointer (tile 1: + 1024 bytes)
MP = 144         # pair: block-maximum pointer (tile 1: + 64 bytes)
MI, MBLK, MC = 148, 150, 152     # + x
T0 = 154         # temporaries T0 .. T0 + 15 (T0 .. T0 + 7 carry wave-uniform inputs until the prologue has read them)
RA = 170         # ring read addresses (16)

S_M0, S_R, S_END, S_LAST, S_LDSK, S_LDSV, S_LDSR = 40, 41, 42, 43, 44, 45, 46
S_RDIST, S_RDW8, S_DLO8 = 47, 48, 49
S_RS = 50        # .. 53: ring offsets of distance tiles D0 .. D3
S_KST, S_VST, S_STEP = 54, 56, 58     # pairs
S_RRS, S_LM1 = 60, 61
S_PST, S_MST = 62, 64                 # pairs
S_C2, S_T0, S_T1, S_JH = 66, 67, 68, 69
S_KB, S_VB, S_RB = 70, 72, 74         # pairs: K / V tile base of the next request, ring source base
S_SUB, S_RET = 76, 80                 # pairs: addresses of the two rescale subroutines (76, 78), return address

OFF_K, OFF_V, OFF_R, TP, TWB = 0, 3 * 8192, 6 * 8192, 68, 4352
THR_EXP2 = float(os.environ.get("FW3_THR", 8.0))


def mfma(dst, a, b, c=None):
    cs = "0" if c is None else R(c, 4)
    rd = set(rng(a, 4)) | set(rng(b, 4)) | (set(rng(c, 4)) if c is not None else set())
    return I(f"v_mfma_f32_16x16x32_bf16 {R(dst, 4)}, {R(a, 4)}, {R(b, 4)}, {cs}", "mfma", rd, rng(dst, 4), srcc=(rng(c, 4) if c is not None else ()))


def valu(op, dst, *src, kind="valu"):
    ops = ", ".join(R(x) if isinstance(x, int) else x for x in src)
    return I(f"{op} {R(dst)}, {ops}", kind, {x for x in src if isinstance(x, int)}, {dst})


def ds_read_b128(dst, addr, off):
    return I(f"ds_read_b128 {R(dst, 4)}, {R(addr)} offset:{off}", "dsr", {addr}, rng(dst, 4))


def ds_read_tr(dst, addr, off):
    return I(f"ds_read_b64_tr_b16 {R(dst, 2)}, {R(addr)} offset:{off}", "dsr", {addr}, rng(dst, 2))


def ds_read_b32(dst, addr, off=0):
    return I(f"ds_read_b32 {R(dst)}, {R(addr)}" + (f" offset:{off}" if off else ""), "dsr", {addr}, {dst})


def ds_write2(addr, d0, d1, o0, o1):
    assert 0 <= o0 < 256 and 0 <= o1 < 256
    return I(f"ds_write2_b32 {R(addr)}, {R(d0)}, {R(d1)} offset0:{o0} offset1:{o1}", "dsw", {addr, d0, d1}, ())


# ---------------------------------------------------------------------------------------------- building blocks
def ptr_steps():
    """K base advances while block r + 3 exists, V base while block r + 2 exists (decided before this iteration's requests)"""
    return [salu(f"s_add_u32 {sr(S_T0)}, {sr(S_R)}, 3"), salu(f"s_cmp_le_u32 {sr(S_T0)}, {sr(S_JH)}"), salu(f"s_cselect_b64 {sr(S_KST, 2)}, {sr(S_STEP, 2)}, 0"),
            salu(f"s_add_u32 {sr(S_T0)}, {sr(S_R)}, 2"), salu(f"s_cmp_le_u32 {sr(S_T0)}, {sr(S_JH)}"), salu(f"s_cselect_b64 {sr(S_VST, 2)}, {sr(S_STEP, 2)}, 0")]


def dma(r6):
    """six LDS-DMA requests of iteration r (r % 6 == r6): two pieces each of K[r + 2], V[r + 1] and the ring rows of block r + 2"""
    kst, vst = (r6 + 2) % 3, (r6 + 1) % 3
    g = []
    for lds, st, base, step in ((S_LDSK, kst, S_KB, S_KST), (S_LDSV, vst, S_VB, S_VST)):
        for p in range(2):
            grp = [salu(f"s_add_u32 m0, {sr(lds)}, {st * 8192 + p * 1024}"), nop(0),
                   I(f"global_load_lds_dwordx4 {R(KOFF + p)}, {sr(base, 2)}", "vmem", {KOFF + p}, ())]
            if p == 1:
                grp += [salu(f"s_add_u32 {sr(base)}, {sr(base)}, {sr(step)}"), salu(f"s_addc_u32 {sr(base + 1)}, {sr(base + 1)}, {sr(step + 1)}")]
            g.append(grp)
    for p in range(2):
        t = T0 + 12 + p
        grp = [valu("v_add_u32", t, sr(S_RDIST), SROW)]
        if p:
            grp.append(valu("v_add_u32", t, "4", t))
        grp += [I(f"v_med3_i32 {R(t)}, {R(t)}, 0, {sr(S_LM1)}", "valu", {t}, {t}),
                I(f"v_mad_u32_u24 {R(t)}, {R(t)}, {sr(S_RRS)}, {R(ROFF + p)}", "valu", {t, ROFF + p}, {t}),
                salu(f"s_add_u32 {sr(S_T0)}, {sr(S_RDW8)}, {p * 0x400}"), salu(f"s_and_b32 {sr(S_T0)}, {sr(S_T0)}, 0xff00"),
                salu(f"s_add_u32 m0, {sr(S_T0)}, {sr(S_LDSR)}"), nop(0),
                I(f"global_load_lds_dwordx4 {R(t)}, {sr(S_RB, 2)}", "vmem", {t}, ())]
        if p:
            grp += [salu(f"s_sub_u32 {sr(S_RDIST)}, {sr(S_RDIST)}, 32"), salu(f"s_sub_u32 {sr(S_RDW8)}, {sr(S_RDW8)}, 0x2000")]
        g.append(grp)
    return g


def band_scalars(ntiles):
    """ring offsets of distance tiles D0 .. of block r from S_DLO8 = (iw - 32 r - 32) << 8, then step it to the next block"""
    o = [salu(f"s_and_b32 {sr(S_RS)}, {sr(S_DLO8)}, 0xff00")]
    for d in range(1, ntiles):
        o += [salu(f"s_add_u32 {sr(S_T0)}, {sr(S_DLO8)}, {d * 0x1000}"), salu(f"s_and_b32 {sr(S_RS + d)}, {sr(S_T0)}, 0xff00")]
    o.append(salu(f"s_sub_u32 {sr(S_DLO8)}, {sr(S_DLO8)}, 0x2000"))
    return o


def ring_reads(ntiles, r6=0):
    """address adds + fragment reads, k-step major (the MFMAs consume them in that order)"""
    ad, rd = [], []
    for ks in range(4):
        for d in range(ntiles):
            ad.append(valu("v_add_u32", RA + 4 * d + ks, sr(S_RS + d), RING + ks))
            rd.append(ds_read_b128(AR(d, ks, r6), RA + 4 * d + ks, OFF_R))
    return ad, rd


def k_reads(stg):
    return [ds_read_b128(AK(t, ks), ROWF + 4 * t + ks, OFF_K + stg * 8192) for ks in range(4) for t in range(2)]


def vt_reads(stg):
    o = []
    for db in range(8):
        o.append(ds_read_tr(AVT(db), TR + db, OFF_V + stg * 8192))
        o.append(ds_read_tr(AVT(db) + 2, TR + 8 + db, OFF_V + stg * 8192))
    return o


def softmax_top(x, cur):
    """deferred running maximum of tile x: the running maximum only moves when the block exceeds it by more than 2^THR (rare: the
    O / row-sum rescale is a subroutine, rescale_sub), then mc = -m c2"""
    thr_hex = "0x%08x" % struct.unpack("<I", struct.pack("<f", THR_EXP2))[0]
    t = T0 + 8 * x
    return [valu("v_sub_f32", t, MBLK + x, MI + x), valu("v_mul_f32", t, sr(S_C2), t),
            I(f"v_cmp_lt_f32 vcc, {thr_hex}, {R(t)}", "valu", {t}, ()),
            salu("s_cmp_eq_u64 vcc, 0"), I(f"s_cbranch_scc1 L_nr{x}_@", "branch"),
            I(f"s_swappc_b64 {sr(S_RET, 2)}, {sr(S_SUB + 2 * x, 2)}", "branch"),
            I(f"L_nr{x}_@:", "label"), I(f"v_mul_f32_e64 {R(MC + x)}, -{R(MI + x)}, {sr(S_C2)}", "valu", {MI + x}, {MC + x})]


def rescale_sub(x):
    """m = max(m, m_blk), alpha = exp2((m_old - m) c2), O and the row sums of tile x (AGPRs: read / scale / write back, six at a time) *= alpha"""
    t, al = T0 + 8 * x, T0 + 8 * x + 1
    o = [I(f"L_resc{x}_%=:", "label"), valu("v_max_f32", t, MI + x, MBLK + x), valu("v_sub_f32", al, MI + x, t), valu("v_mul_f32", al, sr(S_C2), al),
         valu("v_exp_f32", al, al, kind="trans"), valu("v_mov_b32", MI + x, t), nop(1)]
    for i0 in range(0, 36, 6):
        tmp = [T0 + 8 * x + 2 + j for j in range(6)]
        o += [valu("v_accvgpr_read_b32", tmp[j], AO(x) + i0 + j) for j in range(6)]
        o += [valu("v_mul_f32", tmp[j], tmp[j], al) for j in range(6)]
        o += [valu("v_accvgpr_write_b32", AO(x) + i0 + j, tmp[j]) for j in range(6)]
    return o + [nop(1), I(f"s_setpc_b64 {sr(S_RET, 2)}", "branch")]


PACK = os.environ.get("FW3_PACK", "0") != "0"   # packed fp32 VALU (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction, the same roundings).  Measured round 6, same box, B = 64: 311 instead of 327 instructions per iteration, bit-identical results, and 832-836 us against 816-822 us: SLOWER (a packed fp32 op takes two passes; the unpacked stream interleaves finer) -- off, profiles/r06h_flash_fwd3_pack.txt


def exp_pairs(x, cur):
    return [[I(f"v_fma_f32 {R(cur + i)}, {R(cur + i)}, {sr(S_C2)}, {R(MC + x)}", "valu", {cur + i, MC + x}, {cur + i}),
             valu("v_exp_f32", cur + i, cur + i, kind="trans")] for i in range(8)]


def exp_pk(x, cur):
    """(s c2 + mc) of elements 2 j, 2 j + 1 as ONE v_pk_fma_f32: c2 is read twice from the low half of its (even-aligned) SGPR pair, mc of tile
    x from half x of the (MC, MC + 1) register pair (op_sel picks the half for the low result, op_sel_hi for the high one)"""
    sel = f"op_sel:[0,0,{x}] op_sel_hi:[1,0,{x}]"
    return [I(f"v_pk_fma_f32 {R(cur + 2 * j, 2)}, {R(cur + 2 * j, 2)}, {sr(S_C2, 2)}, {R(MC, 2)} {sel}", "valu",
              {cur + 2 * j, cur + 2 * j + 1, MC, MC + 1}, {cur + 2 * j, cur + 2 * j + 1}) for j in range(4)]


def exps(x, cur):
    return [valu("v_exp_f32", cur + i, cur + i, kind="trans") for i in range(8)]


def cvt(x, cur):
    return [valu("v_cvt_pk_bf16_f32", PB(x) + j, cur + 2 * j, cur + 2 * j + 1) for j in range(4)]


def stores():
    """p~ images of both tiles (1 KiB each) and ONE 128-byte store of the 32 block maxima of the wave's queries (lanes 0-15: tile 0,
    16-31: tile 1; the pointer is per lane).  A store costs ~100 issue cycles of the wave: three instead of four."""
    o = [I(f"global_store_dwordx4 {R(PP, 2)}, {R(PB(0), 4)}, off", "vmem", set(rng(PB(0), 4)) | {PP, PP + 1}, ()),
         I(f"global_store_dwordx4 {R(PP, 2)}, {R(PB(1), 4)}, off offset:1024", "vmem", set(rng(PB(1), 4)) | {PP, PP + 1}, ()),
         I(f"v_lshl_add_u64 {R(PP, 2)}, {R(PP, 2)}, 0, {sr(S_PST, 2)}", "valu", {PP, PP + 1}, {PP, PP + 1}),
         salu(f"s_sub_u32 {sr(S_PST)}, {sr(S_PST)}, 2048"),       # the p~ images are a triangle (relattn_flash.h): the row of the next key block is two tiles shorter
         salu("s_mov_b64 exec, 0xffff"),
         I(f"v_mul_f32 {R(T0 + 14)}, {sr(S_C2)}, {R(MI)}", "valu", {MI}, {T0 + 14}),
         salu("s_mov_b64 exec, 0xffff0000"),
         I(f"v_mul_f32 {R(T0 + 14)}, {sr(S_C2)}, {R(MI + 1)}", "valu", {MI + 1}, {T0 + 14}),
         salu("s_mov_b64 exec, 0xffffffff"),
         I(f"global_store_dword {R(MP, 2)}, {R(T0 + 14)}, off", "vmem", {MP, MP + 1, T0 + 14}, ()),
         salu("s_mov_b64 exec, -1"),
         I(f"v_lshl_add_u64 {R(MP, 2)}, {R(MP, 2)}, 0, {sr(S_MST, 2)}", "valu", {MP, MP + 1}, {MP, MP + 1})]
    return o


def scratch_writes(x, acc, col):
    return [ds_write2(TWR + x, acc + 0, acc + 1, col, col + TP), ds_write2(TWR + x, acc + 2, acc + 3, col + 2 * TP, col + 3 * TP)]


def skew_reads(x, par):
    return [ds_read_b32(SK(x) + i, TSK + 8 * par + i, x * TWB) for i in range(8)]


def next_scores(x, nxt):
    if PACK:
        return [I(f"v_pk_add_f32 {R(nxt + 2 * j, 2)}, {R(AS(x) + 2 * j, 2)}, {R(SK(x) + 2 * j, 2)}", "valu",
                  {AS(x) + 2 * j, AS(x) + 2 * j + 1, SK(x) + 2 * j, SK(x) + 2 * j + 1}, {nxt + 2 * j, nxt + 2 * j + 1}) for j in range(4)]
    return [valu("v_add_f32", nxt + i, AS(x) + i, SK(x) + i) for i in range(8)]


def mask_block(x, nxt):
    """diagonal block: key kk(t, g) + r of tile x (16 x rows below tile 0) is visible iff r - 16 x <= thr[t]"""
    o = [salu(f"s_cmp_lg_u32 {sr(S_R)}, {sr(S_LAST)}"), I(f"s_cbranch_scc1 L_nm{x}_@", "branch")]
    rare = []
    for t in range(2):
        for r in range(4):
            rare.append(I(f"v_cmp_le_i32 vcc, {r - 16 * x}, {R(THR + t)}", "valu", {THR + t}, ()))
            rare.append(I(f"v_cndmask_b32 {R(nxt + 4 * t + r)}, {R(NEG)}, {R(nxt + 4 * t + r)}, vcc", "valu", {NEG, nxt + 4 * t + r}, {nxt + 4 * t + r}))
    for z in rare:
        z.rare = True
    return o + rare + [I(f"L_nm{x}_@:", "label")]


def block_max(x, nxt):
    t = T0 + 8 * x
    o = [I(f"v_max3_f32 {R(t)}, {R(nxt)}, {R(nxt + 1)}, {R(nxt + 2)}", "valu", {nxt, nxt + 1, nxt + 2}, {t}),
         I(f"v_max3_f32 {R(t + 1)}, {R(nxt + 3)}, {R(nxt + 4)}, {R(nxt + 5)}", "valu", {nxt + 3, nxt + 4, nxt + 5}, {t + 1}),
         valu("v_max_f32", t + 2, nxt + 6, nxt + 7),
         I(f"v_max3_f32 {R(MBLK + x)}, {R(t)}, {R(t + 1)}, {R(t + 2)}", "valu", {t, t + 1, t + 2}, {MBLK + x})]
    for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
        o += [valu("v_mov_b32", t, MBLK + x),
              I(f"{op} {R(MBLK + x)}, {R(t)}", "perm", {MBLK + x, t}, {MBLK + x, t}),
              valu("v_max_f32", MBLK + x, MBLK + x, t)]
    return o


def tail(nvm, r6):
    return [I(f"s_waitcnt vmcnt({nvm})", "wait"), I("s_barrier", "barrier"),
            salu(f"s_add_u32 {sr(S_R)}, {sr(S_R)}, 1"), salu(f"s_cmp_gt_u32 {sr(S_R)}, {sr(S_END)}"),
            I("s_cbranch_scc1 L_done_%=", "branch"), I(f"s_branch L_inst{(r6 + 1) % 6}_%=", "branch")]


def spread(slots, fillers):
    """fillers: flat list of groups; group i goes behind slot floor(i * n / m)"""
    return interleave(slots, fillers)


# ---------------------------------------------------------------------------------------------- paths
def path_first(r6):
    """r == 0: band tiles D0 .. D3, six (tile, distance-tile) chains, S^T, skewed reads of block 0 (stage 0, parity 0)"""
    assert r6 == 0
    seq = band_scalars(4) + ptr_steps()
    ad, rd = ring_reads(4)
    seq += ad + rd + k_reads(0)
    chains = [(0, 0, T + 0, 0), (0, 1, T + 4, 16), (0, 2, SK(0), 32), (1, 1, T + 8, 0), (1, 2, T + 12, 16), (1, 3, SK(0) + 4, 32)]   # (tile, distance tile, acc, column)
    rel = [mfma(acc, AQV(x, ks), AR(d, ks), None if ks == 0 else acc) for ks in range(4) for (x, d, acc, col) in chains]
    seq += spread(rel, dma(r6))
    smf = [mfma(AS(x, t), AK(t, ks), AQU(x, ks), None if ks == 0 else AS(x, t)) for ks in range(4) for x in range(2) for t in range(2)]
    wr = [scratch_writes(x, acc, col) for (x, d, acc, col) in chains]
    seq += spread(smf, [[z] for g in wr for z in g])
    for x in range(2):
        seq += skew_reads(x, 0)
    for x in range(2):
        seq += next_scores(x, S(x, 0)) + mask_block(x, S(x, 0)) + block_max(x, S(x, 0))
    return seq + tail(6, r6)


def schedule(mf, items, name=""):
    """mf: MFMA instructions (one "slot" each, ~16 cycles of the matrix pipe = four issue cycles of other instructions); items: (earliest
    slot, [instructions]) in priority order.  Every slot takes eligible groups, in order, until it holds its share of what is left."""
    out, pend = [], list(items)
    left = sum(len(g) for _, g in pend)
    counts = []
    for s_, m in enumerate(mf):
        out.append(m)
        quota = -(-left // (len(mf) - s_))
        n, i = 0, 0
        while i < len(pend) and n < quota:
            e, g = pend[i]
            if e <= s_:
                out += g
                n += len(g)
                left -= len(g)
                pend.pop(i)
            else:
                i += 1
        counts.append(n)
    assert not pend, (name, [(e, len(g)) for e, g in pend])
    if os.environ.get("FW3_SHOW"):
        print(name, counts)
    return out


def path_steady(r6):
    """1 <= r <= last: 50 MFMAs (16 relative-term, 16 S^T of block r; 16 P.V + 2 row-sum of block r-1) with everything else spread behind
    them: one wave per SIMD issues one instruction per four cycles, the matrix pipe takes 16 per MFMA, so about three other instructions
    per MFMA hide it -- and the wave is issue-bound, so what counts is that it never stalls (measured first version: 35 % of its cycles)."""
    stg_a, par, stg_b = r6 % 3, r6 % 2, (r6 + 2) % 3
    cur, nxt = 1 - par, par                              # block r's scores go to set par; block r-1's are in the other set
    seq = band_scalars(2)
    ad, rd = ring_reads(2, r6)           # distance tiles D0, D1 of block r; D2 = D0 of block r - 1 is still in its registers (AR)
    seq += ad + rd
    chains = [(0, 0, T + 0, 0), (0, 1, T + 4, 16), (1, 1, T + 8, 0), (1, 2, T + 12, 16)]
    rel = [mfma(acc, AQV(x, ks), AR(d, ks, r6), None if ks == 0 else acc) for ks in range(4) for (x, d, acc, col) in chains]                   # slots 0-15
    smf = [mfma(AS(x, t), AK(t, ks), AQU(x, ks), None if ks == 0 else AS(x, t)) for ks in range(4) for x in range(2) for t in range(2)]      # 16-31
    pv = [mfma(AO(x, db), AVT(db), PB(x), AO(x, db)) for db in range(8) for x in range(2)] + [mfma(AL(x), AONES, PB(x), AL(x)) for x in range(2)]  # 32-49
    items = []
    kr = k_reads(stg_a)
    items += [(0, softmax_top(0, S(0, cur))), (0, kr[0:4]), (1, softmax_top(1, S(1, cur))), (1, kr[4:8]), (2, ptr_steps())]
    # exp2 of block r-1: fma of element k beside exp of element k-1 (the dependent pair is never back to back), p~ pack two groups later
    if PACK:   # pk[m] = elements 2 m, 2 m + 1 (element e = 8 x + i): group 2 m holds pk[m] beside the exp of element 2 m - 1, group 2 m + 1 the exp of 2 m
        pk = [q for x in range(2) for q in exp_pk(x, S(x, cur))]
        ex = [q for x in range(2) for q in exps(x, S(x, cur))]
        exg = [[pk[0]]] + [([pk[k // 2]] if k % 2 == 0 else []) + [ex[k - 1]] for k in range(1, 16)] + [[ex[15]]]
    else:
        fe = [p for x in range(2) for p in exp_pairs(x, S(x, cur))]
        exg = [[fe[0][0]]] + [[fe[k][0], fe[k - 1][1]] for k in range(1, 16)] + [[fe[15][1]]]
    cvs = {x: cvt(x, S(x, cur)) for x in range(2)}
    d = dma(r6)
    for k, grp in enumerate(exg):                                # slots 3 .. 19
        items.append((3 + k, grp))
        if k >= 3 and k % 2 == 1:                                # elements 2j, 2j+1 are exponentiated after group 2j+2: pack behind group 2j+3
            j = (k - 3) // 2
            items.append((3 + k, [cvs[j // 4][j % 4]]))
        if k in (2, 6, 10, 14):                                  # the K / V requests
            items.append((3 + k, d[(k - 2) // 4]))
    items.append((20, [cvs[1][3]]))
    wr = [scratch_writes(x, acc, col ^ (32 * par)) for (x, dd, acc, col) in chains]
    for c in range(4):
        items.append((16 + c, wr[c]))
    skr = skew_reads(0, par) + skew_reads(1, par)
    st = stores()
    vtr = vt_reads(stg_b)
    for j in range(4):
        items.append((20 + j, skr[4 * j:4 * j + 4]))
    ring = []
    for grp in d[4:6]:                                           # ring requests: the address VALU part and the scalar / issue part separately
        k = next(i for i, z in enumerate(grp) if z.kind == "salu")
        ring += [grp[:k], grp[k:]]
    sts = [st[0:1], st[1:4], st[4:11], st[11:12]]               # (the exec window stays in one piece)
    for j in range(8):                                           # V^T fragments of d-block j: two slots ahead of their first MFMA at the latest
        items.append((24 + (3 * j) // 2, vtr[2 * j:2 * j + 2]))
        if j % 2 == 0:
            items.append((24 + (3 * j) // 2, sts[j // 2]))
        else:
            items.append((24 + (3 * j) // 2, ring[j // 2]))
    for x in range(2):
        bm = block_max(x, S(x, nxt))
        ns = next_scores(x, S(x, nxt))
        h = len(ns) // 2
        items += [(35 + 2 * x, ns[0:h]), (36 + 2 * x, ns[h:]), (37 + 2 * x, mask_block(x, S(x, nxt))),
                  (39 + 2 * x, bm[0:4]), (42 + 2 * x, bm[4:7]), (45 + 2 * x, bm[7:10])]
    seq += schedule(rel + smf + pv, items, f"steady{r6}")
    return seq + tail(6 + 3, r6)


def path_last(r6):
    par, stg_b = r6 % 2, (r6 + 2) % 3
    cur = 1 - par
    seq = ptr_steps() + [z for g in dma(r6) for z in g] + vt_reads(stg_b)
    for x in range(2):
        seq += softmax_top(x, S(x, cur))
    for x in range(2):
        if PACK:
            seq += exp_pk(x, S(x, cur)) + exps(x, S(x, cur))
        else:
            for p in exp_pairs(x, S(x, cur)):
                seq += p
    seq += cvt(0, S(0, cur)) + cvt(1, S(1, cur)) + stores()
    for db in range(8):
        for x in range(2):
            seq.append(mfma(AO(x, db), AVT(db), PB(x), AO(x, db)))
    seq += [mfma(AL(x), AONES, PB(x), AL(x)) for x in range(2)]
    return seq + tail(6 + 3, r6)


def path_idle(r6):
    return ptr_steps() + [z for g in dma(r6) for z in g] + tail(6, r6)


def instance(r6):
    out = [I(f"L_inst{r6}_%=:", "label")]
    total = 0

    def add(seq, tag):
        nonlocal total
        fin, pads = finalize(seq, tag)
        total += pads
        for z in fin:
            z.text = z.text.replace("_@", f"_{tag}_%=")
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
    return out, total


def prologue():
    # wave-uniform 32-bit inputs arrive in VGPRs T0 .. T0 + 7 (the asm statement has too few operand slots for them as SGPRs)
    rf = lambda s, i: salu(f"v_readfirstlane_b32 {sr(s)}, {R(T0 + i)}")
    o = [salu(f"s_mov_b32 {sr(S_M0)}, m0"), salu(f"s_mov_b32 {sr(S_R)}, 0"),
         rf(S_JH, 0), rf(S_LAST, 1), rf(S_RDIST, 2), rf(S_RDW8, 3), rf(S_DLO8, 4), rf(S_RRS, 5), rf(S_LM1, 6), rf(S_C2, 7),
         salu(f"s_add_u32 {sr(S_END)}, {sr(S_JH)}, 1"),
         salu(f"s_mov_b32 {sr(S_LDSK)}, %[ldsk]"), salu(f"s_add_u32 {sr(S_LDSV)}, %[ldsk], {OFF_V}"), salu(f"s_mov_b32 {sr(S_LDSR)}, %[ldsr]"),
         salu(f"s_mov_b64 {sr(S_STEP, 2)}, %[kvstep]"), salu(f"s_mov_b64 {sr(S_PST, 2)}, %[ptstep]"),
         salu(f"s_mov_b64 {sr(S_KB, 2)}, %[kbase]"), salu(f"s_mov_b64 {sr(S_VB, 2)}, %[vbase]"), salu(f"s_mov_b64 {sr(S_RB, 2)}, %[rbase]"),
         salu(f"s_add_u32 {sr(S_T0)}, {sr(S_LM1)}, 1"), salu(f"s_lshl_b32 {sr(S_MST)}, {sr(S_T0)}, 2"), salu(f"s_mov_b32 {sr(S_MST + 1)}, 0")]
    for x in range(2):
        o += [salu(f"s_getpc_b64 {sr(S_SUB + 2 * x, 2)}"), I(f"L_pc{x}_%=:", "label"),
              salu(f"s_add_u32 {sr(S_SUB + 2 * x)}, {sr(S_SUB + 2 * x)}, L_resc{x}_%=-L_pc{x}_%="), salu(f"s_addc_u32 {sr(S_SUB + 2 * x + 1)}, {sr(S_SUB + 2 * x + 1)}, 0")]
    o += [I(f"v_mov_b32 {R(T0 + 8)}, 0", "valu"), I(f"v_mov_b32 {R(T0 + 9)}, 0x3f803f80", "valu"), nop(0)]
    for x in range(2):
        o += [I(f"v_accvgpr_write_b32 {R(AO(x) + i)}, {R(T0 + 8)}", "valu") for i in range(36)]
    o += [I(f"v_accvgpr_write_b32 {R(AONES + i)}, {R(T0 + 9)}", "valu") for i in range(4)]
    for x in range(2):
        o += [I(f"v_mov_b32 {R(MI + x)}, 0xf149f2ca", "valu"), I(f"v_mov_b32 {R(MBLK + x)}, 0xf149f2ca", "valu")]
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
    for x in range(2):          # (every path ends in a branch: nothing falls into the subroutines)
        lines += rescale_sub(x)
    lines += epilogue()
    abl = os.environ.get("FW3_ABLATE", "")   # timing-only ablations (results wrong): dma, store, mfma, lds, valu, barrier
    if abl:
        keep_wr = {PP, PP + 1, MP, MP + 1} | set(range(T0 + 12, T0 + 14)) | set(range(RA, RA + 16))

        def keep(z):
            if "dma" in abl and z.text.startswith("global_load_lds"):
                return False
            if "store" in abl and z.text.startswith("global_store"):
                return False
            if "mfma" in abl and z.kind == "mfma":
                return False
            if "lds" in abl and z.kind in ("dsr", "dsw"):
                return False
            if "barrier" in abl and z.kind == "barrier":
                return False
            if "valu" in abl and z.kind in ("valu", "trans", "perm") and z.wr and not (set(z.wr) & keep_wr):
                return False
            return True
        lines = [z for z in lines if keep(z)]
    path = os.path.join(d, "relattn_flash_fwd3_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_flash_fwd3.py -- do not edit; the schedule and the register map are described there.\n")
        for z in lines:
            f.write(f'"{z.text}\\n\\t"\n')
    n_steady = len(finalize(path_steady(1), "x")[0])
    print(f"{path}: {len(lines)} instructions, steady path {n_steady}, hazard pads {pads} states")


if __name__ == "__main__":
    main()
