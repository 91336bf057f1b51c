#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of the 4-wave NT GEMM (bdm_db1_amd/csrc/gemm_w4_loop.inc).

One wave per SIMD owns a 128 x 128 piece of the 256 x 256 workgroup tile (256 accumulator registers in AGPRs), so every
fragment read from LDS feeds 8 MFMAs instead of the 8-wave kernel's 4 / 8 (A / B).  There is no partner wave to hide the
fragment reads and the LDS-DMA issue behind, so the loop is software-pipelined by hand: this script emits the instruction
stream (physical registers, counted waits) as ONE inline-asm string; hipcc's scheduler, waitcnt pass and register allocator
never see it.  Timeline of k-tile t (stage s = t & 1), one "slot" = one MFMA (16 cycles):

  P0  64 MFMAs on fragment set F0 = (t, ks 0)   | slots 0-15: ds_read (t, ks 1) -> F1 | after 31: lgkmcnt(0), barrier A
      (stage s is free now) | slots 33..61: 8 LDS-DMA of k-tile t+2 -> stage s
  P1  64 MFMAs on F1                            | slots 1..29: 8 LDS-DMA | slot 30: pointer step | after 39: vmcnt, barrier B
      (k-tile t+1 has landed) | slots 44-59: ds_read (t+1, ks 0) -> F0 | end: lgkmcnt(0)

Physical registers: a[0:255] accumulators, acc[i][j] = a[4 (8 i + j) ..]; v[0:31] F0.A, v[32:63] F0.B, v[64:95] F1.A,
v[96:127] F1.B; v128-v135 fragment read addresses [stage][operand][ks]; v136-v151 DMA lane offsets [operand][sub-tile][piece];
s[72:73] / s[74:75] A / B pointers of the next k-tile to request, s76-s79 temporaries, s80 loop counter, s81 wrap counter.
Named operands: see gemm_w4.hip.
"""
import sys

FA = {0: 0, 1: 64}     # fragment set -> first A register
FB = {0: 32, 1: 96}
RD = 128               # + stage * 4 + op * 2 + ks
VOFF = 136             # + op * 8 + sub * 4 + it


def mfma(fs, slot):
    i, j = slot >> 3, slot & 7
    c = 4 * (8 * i + j)
    return f"v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], v[{FB[fs] + 4 * j}:{FB[fs] + 4 * j + 3}], v[{FA[fs] + 4 * i}:{FA[fs] + 4 * i + 3}], a[{c}:{c + 3}]"


def frag_read(fs, stage, ks, n):
    """n = 0..15: A fragments 0-7 then B fragments 0-7 of (stage, ks) into set fs"""
    op, f = n >> 3, n & 7
    dst = (FA[fs] if op == 0 else FB[fs]) + 4 * f
    return f"ds_read_b128 v[{dst}:{dst + 3}], v{RD + stage * 4 + op * 2 + ks} offset:{f * 2048}"


def dma(stage, d):
    """d = 0..15: (operand, sub-tile, piece); returns (m0 setup, load)"""
    op, sub, it = d >> 3, (d >> 2) & 1, d & 3
    imm = stage * 65536 + (op * 2 + sub) * 16384 + it * 1024
    ptr = "s[72:73]" if op == 0 else "s[74:75]"
    return (f"s_add_u32 m0, %[ldsw], {imm}", f"global_load_lds_dwordx4 v{VOFF + op * 8 + sub * 4 + it}, {ptr}")


def ptr_step():
    return ["s_sub_u32 s81, s81, 1", "s_cmp_eq_u32 s81, 0", "s_cselect_b32 s76, %[back], 128", "s_cselect_b32 s77, -1, 0",
            "s_add_u32 s72, s72, s76", "s_addc_u32 s73, s73, s77", "s_add_u32 s74, s74, s76", "s_addc_u32 s75, s75, s77"]


def body(stage, do_dma, do_next):
    out = []
    # ---- P0
    for slot in range(64):
        out.append(mfma(0, slot))
        if slot < 16:
            out.append(frag_read(1, stage, 1, slot))
        if slot == 31:
            out += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
        if do_dma and slot >= 32 and slot < 64:
            k, ph = (slot - 32) >> 2, (slot - 32) & 3
            m0set, ld = dma(stage, k)
            if ph == 0:
                out.append(m0set)
            elif ph == 1:
                out.append(ld)
    # ---- P1
    for slot in range(64):
        out.append(mfma(1, slot))
        if do_dma and slot < 32:
            k, ph = 8 + (slot >> 2), slot & 3
            m0set, ld = dma(stage, k)
            if ph == 0:
                out.append(m0set)
            elif ph == 1:
                out.append(ld)
        if do_dma and slot == 32:
            out += ptr_step()
        if do_next and slot == 39:
            out += [f"s_waitcnt vmcnt({16 if do_dma else 0})", "s_barrier"]
        if do_next and 44 <= slot < 60:
            out.append(frag_read(0, stage ^ 1, 0, slot - 44))
    if do_next:
        out.append("s_waitcnt lgkmcnt(0)")
    return out


def prologue():
    out = ["s_mov_b32 s79, m0", "s_mov_b64 s[72:73], %[aptr]", "s_mov_b64 s[74:75], %[bptr]", "s_mov_b32 s80, %[nloops]", "s_mov_b32 s81, %[wrap]"]
    out += ["v_mov_b32 v128, %[rda0]", "v_mov_b32 v129, %[rda1]", "v_mov_b32 v130, %[rdb0]", "v_mov_b32 v131, %[rdb1]"]
    out += [f"v_add_u32 v{132 + k}, 0x10000, v{128 + k}" for k in range(4)]
    for op, (v0, st) in enumerate((("%[voffa]", "%[stepa]"), ("%[voffb]", "%[stepb]"))):
        b = VOFF + op * 8
        out.append(f"v_mov_b32 v{b}, {v0}")
        out += [f"v_add_u32 v{b + it}, {st}, v{b + it - 1}" for it in range(1, 4)]
        out.append(f"s_lshl_b32 s76, {st}, 4")
        out += [f"v_add_u32 v{b + 4 + it}, s76, v{b + it}" for it in range(4)]
    out += [f"v_accvgpr_write_b32 a{n}, 0" for n in range(256)]
    for t in range(2):
        for d in range(16):
            m0set, ld = dma(t, d)
            out += [m0set, "s_nop 0", ld]
        out += ptr_step()
    out += ["s_waitcnt vmcnt(16)", "s_barrier"]
    out += [frag_read(0, 0, 0, n) for n in range(16)]
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def main(path):
    lines = prologue()
    lines.append("L_w4_loop_%=:")
    lines += body(0, True, True) + body(1, True, True)
    lines += ["s_sub_u32 s80, s80, 1", "s_cmp_lg_u32 s80, 0", "s_cbranch_scc1 L_w4_loop_%="]
    lines += body(0, False, True) + body(1, False, False)
    lines += ["s_nop 15", "s_nop 15", "s_mov_b32 m0, s79"]
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4.py -- do not edit; the schedule is described there.\n")
        for ln in lines:
            f.write(f'"{ln}\\n\\t"\n')
    print(f"{path}: {len(lines)} instructions")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bdm_db1_amd/csrc/gemm_w4_loop.inc")
