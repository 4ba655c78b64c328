"""Generator of vist3a_amd/csrc/gemm_w4_loop.inc: the hand-scheduled K loop of `gemm_w4_kernel` (csrc/gemm_bf16.hip) as ONE inline-asm body.

    python tools/gen_gemm_w4.py            (rewrites the .inc; the file is committed, this script documents and reproduces it)

Why assembly: with ONE wave per SIMD (four 64-lane waves per CU, each with the whole 512-entry register file: 192 accumulators in AGPRs) nothing
hides an instruction that stalls the wave's issue - every LDS-DMA, every fragment read and every wait sits in the only MFMA stream there is.  hipcc
neither interleaves memory operations between individual MFMAs on request nor keeps accumulators in AGPRs without shuttling them (DESIGN.md
section 3, round 3), so the loop is written out instruction by instruction: 12 MFMAs per 16-deep k-step, each followed by at most one LDS-DMA
piece and one 16-byte fragment read, fragments double-buffered by k-step, ONE s_barrier per 64-deep K tile.

Tile: 256 (M) x 192 (N) x 64 (K), waves 2 x 2, a wave owns 128 x 96 outputs = 4 x 3 blocks of 32 x 32 (v_mfma_f32_32x32x16_bf16, first operand the
N-side fragment: D^T orientation like every tile of gemm_bf16.hip; per output element the same ascending-k chain: bit-identical results).

Timeline of one K tile u (stage s = u & 1), between barriers B_u and B_(u+1):
    R0: 12 MFMAs of k-step 3 of tile u-1 (fragments F1)   | LDS-DMA pieces A0..A7 of tile u+1 -> stage s^1, reads of k-step 0 (tile u) -> F0
    R1: 12 MFMAs of k-step 0 (F0)                         | pieces B0..B5 of tile u+1, reads of k-step 1 -> F1
    R2: 12 MFMAs of k-step 1 (F1)                         | reads of k-step 2 -> F0, operand base pointers += 128 B, DMA stage toggled
    R3: 12 MFMAs of k-step 2 (F0)                         | reads of k-step 3 -> F1, read addresses toggled to the other stage
    s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier              (tile u+1 landed for everybody; everybody is done reading tile u)
so the matrix pipe has the 12 MFMAs of R0 queued right behind every barrier (their fragments are already in registers), the pieces of the next
tile get two k-steps (~770 matrix-pipe cycles) to land, and a fragment is always read a whole k-step before the MFMAs that use it.

Registers (physical, named in the text): a[0:191] accumulators, block (i, j) at a[(3 i + j) 16 ...]; v[0:31] A fragments (2 buffers x 4 blocks),
v[32:55] B fragments (2 x 3), v56-59 / v60-63 A / B read addresses per k-step, v64-71 / v72-77 per-lane global byte offsets of the 8 A / 6 B
pieces, v78 scratch; s[60:61] / s[62:63] A / B bases (advance 128 B per K tile), s64 the wave's LDS-DMA base of the stage being filled, s65 loop count.
v0-v191 are the statement's (early-clobber) outputs: free as scratch inside the loop, loaded from the AGPRs at the very end."""
import os
import sys
from pathlib import Path

OUT = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "vist3a_amd" / "csrc" / "gemm_w4_loop.inc"
# development ablations of the steady-state loop bodies (W4_ABL=<mask> python tools/gen_gemm_w4.py /tmp/x.inc; results are garbage, timing only):
# 1 = no LDS-DMA pieces, 2 = no vmcnt wait in front of the barrier, 4 = no fragment reads, 8 = no barrier
ABL = int(os.environ.get("W4_ABL", "0"))

# operand numbers of the asm statement (outputs 0..11 = acc blocks; inputs follow) - must match gemm_w4_kernel in gemm_bf16.hip
OPS = dict(sA=12, sB=13, vrowA=14, vrowB=15, vchunk=16, sM1=17, sN1=18, slda2=19, sldb2=20, snk=21, sdma=22, vK0=23, vK1=24, vK2=25, vK3=26,
           sAoff=27, sBoff=28)
STAGE_XOR = 0x10000      # stage 1 = stage 0 + 64 KiB: one xor toggles an address
B_ROWS_OFF = 256 * 128   # LDS offset of the B rows inside a stage

lines = []


def e(s, c=""):
    lines.append((s, c))


def op(n):
    return f"%{OPS[n]}"


def FA(buf, i):
    b = buf * 16 + i * 4
    return f"v[{b}:{b + 3}]"


def FB(buf, j):
    b = 32 + buf * 12 + j * 4
    return f"v[{b}:{b + 3}]"


def acc(i, j):
    b = (i * 3 + j) * 16
    return f"a[{b}:{b + 15}]"


PIECES = [("A", j) for j in range(8)] + [("B", j) for j in range(6)]


def piece_lds_off(p):
    k, j = p
    return j * 4096 + (B_ROWS_OFF if k == "B" else 0)


def m0_for(p):
    return (f"s_add_u32 m0, s64, {piece_lds_off(p)}", f"M0 = LDS destination of piece {p[0]}{p[1]}")


def dma(p):
    k, j = p
    return (f"global_load_lds_dwordx4 v{(64 if k == 'A' else 72) + j}, s[{'60:61' if k == 'A' else '62:63'}]", f"piece {k}{j}: 8 rows x 128 B")


def reads(ks, buf):
    r = [(f"ds_read_b128 {FA(buf, i)}, v{56 + ks} offset:{i * 4096}", f"A block {i}, k-step {ks} -> F{buf}") for i in range(4)]
    r += [(f"ds_read_b128 {FB(buf, j)}, v{60 + ks} offset:{j * 4096}", f"B block {j}, k-step {ks} -> F{buf}") for j in range(3)]
    return r


def region(buf, fillers, first=False, note=""):
    """12 MFMAs on fragment buffer `buf`; fillers[k] = instructions placed behind MFMA k"""
    e("", note)
    k = 0
    for i in range(4):
        for j in range(3):
            c = "0" if first else acc(i, j)
            e(f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {FB(buf, j)}, {FA(buf, i)}, {c}")
            for ins in fillers.get(k, []):
                e(*ins)
            k += 1


def advance_and_toggle():
    return [("s_add_u32 s60, s60, 128", "A base: next K tile"), ("s_addc_u32 s61, s61, 0", ""), ("s_add_u32 s62, s62, 128", "B base"),
            ("s_addc_u32 s63, s63, 0", ""), (f"s_xor_b32 s64, s64, {STAGE_XOR}", "the next tile's pieces go to the other stage")]


def toggle_reads():
    return [(f"v_xor_b32 v{56 + k}, {STAGE_XOR}, v{56 + k}", "fragment reads move to the other stage") for k in range(8)]


def body(with_dma, label):
    """one steady-state K tile (see the module docstring)"""
    e(f"{label}_%=:")
    A = [("A", j) for j in range(8)]
    B = [("B", j) for j in range(6)]
    slotsA = [0, 1, 3, 4, 6, 7, 9, 10]      # two pieces, one free slot, ... : 8 pieces behind the 12 MFMAs of R0
    slotsB = [0, 1, 3, 4, 6, 7]             # 6 behind the first 8 MFMAs of R1
    f0, f1 = {k: [] for k in range(12)}, {k: [] for k in range(12)}
    rd0, rd1 = reads(0, 0), reads(1, 1)
    if with_dma:
        # M0 of a piece is written right behind the PREVIOUS piece (hipcc's own pattern): the MFMA in between is the wait state the SALU
        # write of M0 needs before an LDS-DMA reads it.  The first piece's M0 is written before R0's first MFMA.
        e(*m0_for(A[0]))
        order = [(f0, s, p) for s, p in zip(slotsA, A)] + [(f1, s, p) for s, p in zip(slotsB, B)]
        for idx, (f, s_, p) in enumerate(order):
            if ABL & 1:
                break
            f[s_].append(dma(p))
            if idx + 1 < len(order):
                f[s_].append(m0_for(order[idx + 1][2]))
    for k in range(7):
        if ABL & 4:
            break
        f0[k].append(rd0[k])
        f1[k].append(rd1[k])
    region(1, f0, note=f"R0: k-step 3 of the previous tile (F1) | {'pieces A0-A7 of the next tile, ' if with_dma else ''}reads of k-step 0 -> F0")
    e("s_waitcnt lgkmcnt(0)")
    region(0, f1, note=f"R1: k-step 0 (F0) | {'pieces B0-B5, ' if with_dma else ''}reads of k-step 1 -> F1")
    e("s_waitcnt lgkmcnt(0)")
    f2 = {k: ([] if ABL & 4 else [reads(2, 0)[k]]) for k in range(7)}
    if with_dma:
        adv = advance_and_toggle()
        f2[8], f2[9], f2[10] = adv[0:2], adv[2:4], adv[4:5]
    region(1, f2, note="R2: k-step 1 (F1) | reads of k-step 2 -> F0" + (", operand bases += 128 B, DMA stage toggled" if with_dma else ""))
    e("s_waitcnt lgkmcnt(0)")
    f3 = {k: ([] if ABL & 4 else [reads(3, 1)[k]]) for k in range(7)}
    tg = toggle_reads()
    for k in range(4):
        f3[7 + k] = tg[2 * k: 2 * k + 2]
    if with_dma:
        f3[11] = [("s_sub_u32 s65, s65, 1", "tiles left for this loop")]
    region(0, f3, note="R3: k-step 2 (F0) | reads of k-step 3 -> F1, read addresses -> other stage")
    e("s_waitcnt lgkmcnt(0)" if ABL & 2 else "s_waitcnt vmcnt(0) lgkmcnt(0)", "this wave's pieces of the next tile have landed; its last fragment reads have returned")
    if not ABL & 8:
        e("s_barrier", "... for every wave: the next tile is complete, the stage just read may be refilled")


def generate():
    e("", "---- setup: scratch registers from the operands")
    e(f"s_mov_b64 s[60:61], {op('sA')}")
    e(f"s_mov_b64 s[62:63], {op('sB')}")
    e(f"s_mov_b32 s64, {op('sdma')}")
    for j in range(8):
        e(f"v_add_u32 v78, {32 * j}, {op('vrowA')}", f"row of this lane in piece A{j}")
        e(f"v_min_u32 v78, v78, {op('sM1')}", "clamped at the last row (ragged M: duplicates, never stored)")
        e(f"v_mad_u32_u24 v{64 + j}, v78, {op('slda2')}, {op('vchunk')}", "byte offset = row * lda * 2 + swizzled 16-byte chunk")
    for j in range(6):
        e(f"v_add_u32 v78, {32 * j}, {op('vrowB')}")
        e(f"v_min_u32 v78, v78, {op('sN1')}")
        e(f"v_mad_u32_u24 v{72 + j}, v78, {op('sldb2')}, {op('vchunk')}")
    for k in range(4):
        e(f"v_add_u32 v{56 + k}, {op('sAoff')}, {op('vK' + str(k))}", f"A fragment read address, k-step {k} (stage 0)")
        e(f"v_add_u32 v{60 + k}, {op('sBoff')}, {op('vK' + str(k))}", f"B fragment read address, k-step {k}")
    e("", "---- prologue: tile 0 -> stage 0")
    for p in PIECES:
        e(*m0_for(p))
        e("s_nop 0")
        e(*dma(p))
    for ins in advance_and_toggle():
        e(*ins)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier", "B_0: tile 0 visible to all waves")
    e(f"s_cmp_lt_i32 {op('snk')}, 2")
    e("s_cbranch_scc1 P_NO1_%=")
    e("", "tile 1 -> stage 1 (in flight under the whole first tile)")
    for p in PIECES:
        e(*m0_for(p))
        e("s_nop 0")
        e(*dma(p))
    for ins in advance_and_toggle():
        e(*ins)
    e("P_NO1_%=:")
    for r in reads(0, 0):
        e(*r)
    e("s_waitcnt lgkmcnt(0)")
    region(0, {k: [reads(1, 1)[k]] for k in range(7)}, first=True, note="tile 0, k-step 0: accumulators start from the constant 0")
    e("s_waitcnt lgkmcnt(0)")
    region(1, {k: [reads(2, 0)[k]] for k in range(7)}, note="tile 0, k-step 1")
    e("s_waitcnt lgkmcnt(0)")
    f3 = {k: [reads(3, 1)[k]] for k in range(7)}
    tg = toggle_reads()
    for k in range(4):
        f3[7 + k] = tg[2 * k: 2 * k + 2]
    region(0, f3, note="tile 0, k-step 2")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier", "B_1")
    e(f"s_cmp_lt_i32 {op('snk')}, 2")
    e("s_cbranch_scc1 EPI_%=", "K = 64: a single tile")
    e(f"s_sub_u32 s65, {op('snk')}, 2", "tiles 1 .. nk-2 fetch a successor")
    e("s_cmp_eq_u32 s65, 0")
    e("s_cbranch_scc1 LAST_%=")
    body(True, "LOOP")
    e("s_cmp_lg_u32 s65, 0")
    e("s_cbranch_scc1 LOOP_%=")
    body(False, "LAST")
    e("EPI_%=:")
    region(1, {}, note="k-step 3 of the last tile (F1)")
    e("s_nop 15", "the last MFMAs retire before their accumulators are read")
    e("s_nop 15")
    e("s_nop 15")
    for n in range(192):
        e(f"v_accvgpr_read_b32 v{n}, a{n}")


def main():
    generate()
    out = ["// GENERATED by tools/gen_gemm_w4.py - do not edit; the schedule is documented there.\n"]
    for s, c in lines:
        if not s:
            out.append(f"    /* {c} */\n")
        else:
            out.append(f'    "{s}\\n"' + (f"   /* {c} */" if c else "") + "\n")
    OUT.write_text("".join(out))
    n = sum(1 for s, _ in lines if s and not s.endswith(":"))
    print(f"wrote {OUT} ({n} instructions)")


if __name__ == "__main__":
    main()
