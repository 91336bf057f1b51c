#!/usr/bin/env python3
"""Generator of the hand-scheduled main loops of the 4-wave GEMM (bdm_db1_amd/csrc/gemm_w4_loop_{nt,nn,tn}.inc).

One wave per SIMD owns a 128 x 128 piece of the 256 x 256 workgroup tile (256 accumulator registers in AGPRs), so every
fragment read from LDS feeds 8 MFMAs instead of the 8-wave kernel's 4 / 8 (A / B).  There is no partner wave to hide the
fragment reads and the LDS-DMA issue behind, so the loop is software-pipelined by hand: this script emits the instruction
stream (physical registers, counted waits) as ONE inline-asm string; hipcc's scheduler, waitcnt pass and register allocator
never see it.  Timeline of k-tile t (stage s = t & 1), one "slot" = one MFMA (16 cycles):

  P0  64 MFMAs on fragment set F0 = (t, ks 0)   | slots 0-15: fragment reads (t, ks 1) -> F1 | after 19: lgkmcnt(0), barrier A
      (stage s is free now) | from slot 20, every 7th slot until slot 62 of P1: one of the 16 LDS-DMA requests of k-tile t+2 -> stage s
  P1  64 MFMAs on F1                            | after 39: counted vmcnt (k-tile t+1 has landed, 12 requests of t+2 may be in flight),
      barrier B | slots 40-55: fragment reads (t+1, ks 0) -> F0 | after the last request: pointer step | end: lgkmcnt(0)

Operand layouts (gemm_tile.h): K-major = memory [row][k], LDS image [128 rows][128 B], one ds_read_b128 per fragment, fragment f at
+f * 2048, ks 1 at address ^ 64;  M-major = memory [k][row], LDS image [64 k][256 B], two ds_read_b64_tr_b16 per fragment (+1024 for
the second), fragment f at address ^ (f << 5), ks 1 at +8192.  (The XOR forms need the LDS base 256-byte aligned: checked in the kernel.)

Physical registers: a[0:255] accumulators, acc[i][j] = a[4 (8 i + j) ..]; v[0:31] F0.A, v[32:63] F0.B, v[64:95] F1.A,
v[96:127] F1.B; v128-v143 A / v144-v159 B fragment read addresses (K-major: [stage][ks], M-major: [stage][fragment]);
v160-v175 DMA lane offsets [operand][sub-tile][piece]; s[72:73] / s[74:75] A / B pointers of the next k-tile to request,
s76-s77 / s82-s83 temporaries, s79 saved M0, s80 loop counter, s81 wrap counter.  Named operands: see gemm_w4.hip.
"""
import os
import sys

FA = {0: 0, 1: 64}     # fragment set -> first A register
FB = {0: 32, 1: 96}
RD = (128, 144)        # per operand
VOFF = 160             # + op * 8 + sub * 4 + it


def EARLY():
    return int(os.environ.get("W4_EARLY", 0))


class Gen:
    """nj = 8: 256 x 256 workgroup tile (wave tile 128 x 128).  nj = 4: 256 x 128 (wave tile 128 x 64: the two waves of a tile row share ONE
    128-column B sub-tile and take its fragments 0-3 / 4-7) -- for outputs whose 256 x 256 tiling leaves half the CUs idle (micro-batches of
    4 sequences: o_net, ff2, the data gradients).  Same LDS image, same register map, same accumulator numbering a[4 (8 i + j)] (j < nj): the
    k-tile is 8 nj MFMAs per fragment set instead of 64, 8 + nj fragment reads, 8 + nj LDS-DMA requests."""

    def __init__(self, a_kmajor, b_kmajor, nj=8):
        self.km = (a_kmajor, b_kmajor)
        self.nj = nj
        self.ns = 8 * nj          # MFMA slots per fragment set
        self.nr = 8 + nj          # fragment reads per set
        self.nd = 8 + nj          # LDS-DMA requests per k-tile

    def mfma(self, fs, slot):
        i, j = slot // self.nj, slot % self.nj
        c = 4 * (8 * i + j)
        return f"v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], v[{FB[fs] + 4 * j}:{FB[fs] + 4 * j + 3}], v[{FA[fs] + 4 * i}:{FA[fs] + 4 * i + 3}], a[{c}:{c + 3}]"

    def frag_read(self, fs, stage, ks, n):
        """n = 0 .. 7 + nj: A fragments 0-7 then B fragments 0 .. nj - 1 of (stage, ks) into set fs; 1 or 2 instructions"""
        op, f = (0, n) if n < 8 else (1, n - 8)
        dst = (FA[fs] if op == 0 else FB[fs]) + 4 * f
        if self.km[op]:
            return [f"ds_read_b128 v[{dst}:{dst + 3}], v{RD[op] + stage * 2 + ks} offset:{f * 2048}"]
        a = RD[op] + stage * 8 + f
        return [f"ds_read_b64_tr_b16 v[{dst}:{dst + 1}], v{a} offset:{ks * 8192}", f"ds_read_b64_tr_b16 v[{dst + 2}:{dst + 3}], v{a} offset:{ks * 8192 + 1024}"]

    def dma(self, stage, d):
        """d = 0 .. 7 + nj: (operand, sub-tile, piece): A's two sub-tiles, then B's two (nj = 8) or one (nj = 4); returns (m0 setup, load)"""
        op, sub, it = d >> 3, (d >> 2) & 1, d & 3
        imm = stage * 65536 + (op * 2 + sub) * 16384 + it * 1024
        ptr = "s[72:73]" if op == 0 else "s[74:75]"
        return (f"s_add_u32 m0, %[ldsw], {imm}", f"global_load_lds_dwordx4 v{VOFF + op * 8 + sub * 4 + it}, {ptr}")

    def ptr_step(self):
        # (s81 counts the k-tiles until the next "wrap" step and is reloaded from %[wrapn] there: the per-XCD k rotation wraps once -- wrapn
        #  is then beyond the loop's length --, the structural-zero mode walks PERIODS of k: wrapn k-tiles, then a longer step over the skipped ones)
        return ["s_sub_u32 s81, s81, 1", "s_cmp_eq_u32 s81, 0",   # (all selects before the adds rewrite scc; 64-bit steps: one pass over k can exceed 2 GiB)
                "s_cselect_b64 s[76:77], %[backa], %[stepka]", "s_cselect_b64 s[82:83], %[backb], %[stepkb]", "s_cselect_b32 s81, %[wrapn], s81",
                "s_add_u32 s72, s72, s76", "s_addc_u32 s73, s73, s77", "s_add_u32 s74, s74, s82", "s_addc_u32 s75, s75, s83"]

    def body(self, stage, do_dma, do_next, woff=0):
        # slots of barrier A, barrier B, the first read of the next k-tile, and the spacing of the 16 LDS-DMA requests.  Measured on the twelve
        # in-step GEMM shapes (tools/exp/exp_w4.py, sum of their times): A 31 / spacing 4 / B 39 / reads from 44: 12.21 ms; A 19: 12.09;
        # spacing 5 / 6 / 7: 11.92 / 11.80 / 11.53 (2 or 3: 13.1 -- the requests must not queue up); reads from 40: 11.43; A 17 or 23,
        # B 35: worse; B 43: same; W4_EARLY=7 (seven A fragments of the next ks-1 set read at the end of the previous half, barrier A after
        # slot 13-19): 11.48-11.71, no gain.  (W4_* environment variables: for such experiments only.)
        if self.nj == 8:
            A, B, RD = (int(os.environ.get(k, d)) for k, d in (("W4_A", 19), ("W4_B", 39), ("W4_RD", 40)))
            SPF = float(os.environ.get("W4_SP", 7))
        else:   # 32 slots per fragment set: the same order of events at half the distances (12 reads, 12 requests per k-tile)
            A, B, RD = (int(os.environ.get(k, d)) for k, d in (("W4N_A", 13), ("W4N_B", 18), ("W4N_RD", 19)))
            SPF = float(os.environ.get("W4N_SP", 4))
        NS, NR, ND = self.ns, self.nr, self.nd
        # DMA d: M0 setup after global slot A + 1 + SP * d, load one slot later (global slot = P0 slot, or NS + P1 slot)
        ev = {}
        if do_dma:
            for d in range(ND):
                g = A + 1 + int(SPF * d) + woff
                ev.setdefault(g, []).append(self.dma(stage, d)[0])
                ev.setdefault(g + 1, []).append(self.dma(stage, d)[1])
            ev.setdefault(A + 1 + int(SPF * (ND - 1)) + woff + 1, []).extend(self.ptr_step())
            assert A + 1 + int(SPF * (ND - 1)) + woff + 1 < 2 * NS
        before_b = sum(1 for d in range(ND) if A + 1 + int(SPF * d) + woff + 1 <= NS + B) if do_dma else 0   # loads of k-tile t+2 already issued at barrier B
        E = EARLY() if self.nj == 8 else 0
        assert RD + NR <= NS
        out = []
        for slot in range(NS):  # ---- P0
            out.append(self.mfma(0, slot))
            if slot < NR - E:
                out += self.frag_read(1, stage, 1, slot + E)
            if slot == A:
                out += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
            out += ev.get(slot, [])
        pending = 0
        for slot in range(NS):  # ---- P1
            out.append(self.mfma(1, slot))
            out += ev.get(NS + slot, [])
            if do_next and slot == B:
                out += [f"s_waitcnt vmcnt({before_b})", "s_barrier"]
            if do_next and RD <= slot < RD + NR:
                out += self.frag_read(0, stage ^ 1, 0, slot - RD)
            if do_next and RD + NR <= slot < RD + NR + E:   # the first E A-fragments of (t+1, ks 1): A fragment i of F1 was last used in slot 8 i + 7
                i = slot - RD - NR
                assert slot >= 8 * i + 8
                r = self.frag_read(1, stage ^ 1, 1, i)
                pending += len(r)
                out += r
        if do_next:
            out.append(f"s_waitcnt lgkmcnt({pending})")
        return out

    def prologue(self):
        out = ["s_mov_b32 s79, m0", "s_mov_b64 s[72:73], %[aptr]", "s_mov_b64 s[74:75], %[bptr]", "s_mov_b32 s80, %[nloops]", "s_mov_b32 s81, %[wrap]"]
        for op, rd in enumerate(("%[rda]", "%[rdb]")):
            b = RD[op]
            if self.km[op]:   # [stage][ks]
                out += [f"v_mov_b32 v{b}, {rd}", f"v_xor_b32 v{b + 1}, 64, v{b}", f"v_add_u32 v{b + 2}, 0x10000, v{b}", f"v_add_u32 v{b + 3}, 0x10000, v{b + 1}"]
            else:             # [stage][fragment]
                out.append(f"v_mov_b32 v{b}, {rd}")
                out += [f"v_xor_b32 v{b + f}, {f << 5}, v{b}" for f in range(1, 8)]
                out += [f"v_add_u32 v{b + 8 + f}, 0x10000, v{b + f}" for f in range(8)]
        for op, (v0, st, sub) in enumerate((("%[voffa]", "%[pstepa]", "%[suba]"), ("%[voffb]", "%[pstepb]", "%[subb]"))):
            b = VOFF + op * 8
            out.append(f"v_mov_b32 v{b}, {v0}")
            for it in range(1, 4):
                out.append(f"v_add_u32 v{b + it}, {st}, v{b + it - 1}")
                if not self.km[op] and it == 2:
                    out.append(f"v_xor_b32 v{b + it}, 0x80, v{b + it}")   # k-rows 8..15 of a 16-row group: chunk index ^ 8
            if op == 1 and self.nj == 4:
                continue                                                   # one B sub-tile only
            if self.km[op]:
                # the second sub-tile of 128 rows: byte distance as an operand (normally 16 pieces of 8 rows = 128 rows further on; the
                # GEGLU-epilogue NT kernel interleaves value and gate rows of W and places its sub-tiles 64 rows apart, gemm_w4.hip)
                out += [f"v_add_u32 v{b + 4 + it}, {sub}, v{b + it}" for it in range(4)]
            else:
                out += [f"v_add_u32 v{b + 4 + it}, 0x100, v{b + it}" for it in range(4)]   # 128 rows = 256 bytes further along the k-row
        for t in range(2):
            for d in range(self.nd):
                m0set, ld = self.dma(t, d)
                out += [m0set, "s_nop 0", ld]
            out += self.ptr_step()
        out += [f"v_accvgpr_write_b32 a{4 * (8 * i + j) + r}, 0" for i in range(8) for j in range(self.nj) for r in range(4)]   # (while the first two k-tiles are on their way)
        out += [f"s_waitcnt vmcnt({self.nd})", "s_barrier"]
        for n in range(self.nr):
            out += self.frag_read(0, 0, 0, n)
        for n in range(EARLY()):
            out += self.frag_read(1, 0, 1, n)
        out.append("s_waitcnt lgkmcnt(0)")
        return out

    def emit(self, path):
        lines = self.prologue()
        lines.append("L_w4_loop_%=:")
        lines += self.body(0, True, True) + self.body(1, True, True)
        lines += ["s_sub_u32 s80, s80, 1", "s_cmp_lg_u32 s80, 0", "s_cbranch_scc1 L_w4_loop_%="]
        lines += self.body(0, False, True) + self.body(1, False, False)
        lines += ["s_nop 15", "s_nop 15", "s_mov_b32 m0, s79"]
        abl = os.environ.get("W4_ABLATE", "")   # timing / power ablations of the loop (results wrong): dma, reads, barriers, mfma
        if abl:
            pro = len(self.prologue())

            def keep(i, ln):
                if i < pro:
                    return True
                if "dma" in abl and (ln.startswith("global_load_lds") or ln.startswith("s_add_u32 m0")):
                    return False
                if "reads" in abl and ln.startswith("ds_read"):
                    return False
                if "barriers" in abl and ln == "s_barrier":
                    return False
                if "mfma" in abl and ln.startswith("v_mfma") and (int(ln.split("a[")[1].split(":")[0]) // 4) % 8 != 0:
                    return False
                return True
            lines = [ln for i, ln in enumerate(lines) if keep(i, ln)]
            if "dma" in abl:
                lines = [("s_waitcnt vmcnt(0)" if ln.startswith("s_waitcnt vmcnt") else ln) for ln in lines]
        with open(path, "w") as f:
            f.write("// GENERATED by tools/gen_gemm_w4.py -- do not edit; the schedule is described there.\n")
            for ln in lines:
                f.write(f'"{ln}\\n\\t"\n')
        print(f"{path}: {len(lines)} instructions")


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bdm_db1_amd", "csrc")
    for name, ak, bk in (("nt", True, True), ("nn", True, False), ("tn", False, False)):
        Gen(ak, bk).emit(os.path.join(d, f"gemm_w4_loop_{name}.inc"))
        Gen(ak, bk, nj=4).emit(os.path.join(d, f"gemm_w4n_loop_{name}.inc"))
