#!/usr/bin/env python3
"""Writes video-quierer_amd/csrc/gemm_asm256_loop.inc: the K loop of gemm_tn256a_kernel (gemm_asm256.h) as ONE inline-asm
text with fixed registers, scheduled by hand (by this script) instead of by the compiler.

Geometry (gemm_asm256.h): 256x256x64 tile, FOUR waves = one per SIMD, 2 (m) x 2 (n), 128 x 128 per wave = 8 x 8 MFMA
16x16x32 tiles = 256 accumulator registers in a[0:255]; two 64-KiB K-tile buffers in LDS with the image of
gemm_mfma256.h (row r of A at r*128, of W at 32768 + r*128, 16-byte chunks XOR-swizzled by (r>>1)&7).

A K-tile is two halves of 64 MFMAs (k = 0..31 and 32..63 of the tile).  Fragment registers are double-buffered by half:
    WF[h][j] = v[128 + 64h + 4j ..+3]   (W rows 16j..16j+15 of the wave, k-half h)
    AF[h][i] = v[160 + 64h + 4i ..+3]   (A rows 16i..16i+15)
    acc(i,j) = a[(8i + j)*4 ..+3]       += WF[h][j] (srcA) x AF[h][i] (srcB): a lane holds 4 consecutive n of one m
Per tile t in buffer p = t & 1, between the MFMAs ("slot s" = behind the s-th MFMA of the half):
    half 0   16 ds_read_b128: k-half 1 of tile t (buffer p) -> F[1];  lgkmcnt(0), barrier X: buffer p is dead
             LDS-DMA of tile t+2 -> buffer p starts (8 A pieces, then 8 W pieces)
    half 1   the DMA continues;  vmcnt(n), barrier Y: tile t+1 has landed for every wave (n = pieces of tile t+2 issued so far)
             16 ds_read_b128: k-half 0 of tile t+1 (buffer p^1) -> F[0];  lgkmcnt(0) at the end
A piece of LDS-DMA = 8 rows x 128 B = 1 KiB = one buffer_load_dwordx4 ... lds; wave w brings rows 64w..64w+63 of A and of W
(8 + 8 pieces per K-tile); M0 carries the LDS address, the row-group offsets sit in SGPRs, the K offset advances the
descriptors' base.

The four waves run in lockstep (two barriers per K-tile), one per SIMD, and share the CU's one address path: with the same
text on every wave all four issue a piece in the same 16-cycle slot, and each waits for the others' - measured (ablations
below, 8192^3): MFMAs alone 698 us, DMA alone 535-755 us, both 890-1015 us, i.e. 25-46 cycles of MFMA issue lost per piece.
So the text exists in FOUR variants, one per wave: wave w issues its pieces in slots = w (mod `dma_stride` = 4), one piece
per slot over the whole CU.  (`same_slots=1` gives every wave variant 0's slots: the A/B.)

Fixed registers (all on the clobber list; the compiler keeps its own values elsewhere):
    v116..v119  fragment read addresses, buffer 0: A k-half 0, A k-half 1, W k-half 0, W k-half 1;  v120..v123 buffer 1
    v124..v127  DMA lane offsets: A even pieces, A odd pieces (^64), W even, W odd
    v128..v255  fragments;  a0..a255 accumulators
    s[56:59] / s[60:63]  buffer descriptors of A / W (base advanced by 128 B per K-tile)
    s40..s46 / s47..s53  q * 8 rows * lda (ldw) * 2 bytes, q = 1..7
    s54  M0 of the wave's first A piece in buffer 0;  s55  loop counter
"""
import sys, os

MFMA = 'MFMA'     # placeholder: "v_mfma_f32_16x16x32_" TY
WF = lambda h, j: 128 + 64 * h + 4 * j
AF = lambda h, i: 160 + 64 * h + 4 * i
RD_A = lambda buf, h: 116 + 4 * buf + h
RD_W = lambda buf, h: 118 + 4 * buf + h
V_A = (124, 125)
V_W = (126, 127)
S_ROW_A = lambda q: 39 + q        # q = 1..7
S_ROW_W = lambda q: 46 + q
BUF_BYTES, W_REGION, PIECE = 65536, 32768, 1024


def vr(base):
    return f'v[{base}:{base + 3}]'


def mfma(h, i, j):
    c = (8 * i + j) * 4
    return f'{MFMA} a[{c}:{c + 3}], {vr(WF(h, j))}, {vr(AF(h, i))}, a[{c}:{c + 3}]'


def reads_into(h, buf):
    """16 ds_read_b128 filling F[h] from buffer `buf`: the 8 W fragments first (all needed by the half's first MFMAs)."""
    out = [f'ds_read_b128 {vr(WF(h, j))}, v{RD_W(buf, h)} offset:{j * 2048}' for j in range(8)]
    out += [f'ds_read_b128 {vr(AF(h, i))}, v{RD_A(buf, h)} offset:{i * 2048}' for i in range(8)]
    return out


def dma_pieces(buf):
    """The 16 pieces of one K-tile into buffer `buf`: (m0 setup to place one slot earlier or None, [piece + what follows it])."""
    out = []
    for srd, vv, srow, region in ((56, V_A, S_ROW_A, 0), (60, V_W, S_ROW_W, W_REGION)):
        for q in range(8):
            soff = '0' if q == 0 else f's{srow(q)}'
            g = [f'buffer_load_dwordx4 v{vv[q & 1]}, s[{srd}:{srd + 3}], {soff} offen lds']
            if q < 7:
                g.append(f's_add_u32 m0, m0, {PIECE}')
            else:
                g += [f's_add_u32 s{srd}, s{srd}, 128', f's_addc_u32 s{srd + 1}, s{srd + 1}, 0']
                if region == 0:
                    g.append(f's_add_u32 m0, s54, {buf * BUF_BYTES + W_REGION}')
            out.append((f's_add_u32 m0, s54, {buf * BUF_BYTES}' if (region == 0 and q == 0) else None, g))
    return out


def tile(p, kind, sched, w):
    """One K-tile in buffer p for wave w.  kind: 'steady' (DMA of tile t+2, reads of tile t+1), 'pen' (no DMA; tile t+1 is the
    last), 'last' (nothing behind it)."""
    extra = [[] for _ in range(128)]          # slot s of half h -> extra[64 h + s]
    no_dma, no_reads = sched.get('no_dma', 0), sched.get('no_reads', 0)

    def put(slot, ins):
        extra[slot] += ins if isinstance(ins, list) else [ins]

    # half 0: k-half 1 of this tile -> F[1]
    for n, ins in enumerate(reads_into(1, p)):
        if not no_reads:
            put(sched['rd0_first'] + n * sched['rd0_stride'], ins)
    put(sched['x_wait'] if kind != 'last' else 60, 's_waitcnt lgkmcnt(0)')
    if kind == 'steady':
        put(sched['x_bar'], 's_barrier')
        first = sched['dma_first'] + (0 if sched.get('same_slots', 0) else w)
        for n, (setup, g) in enumerate(dma_pieces(p)):
            slot = first + n * sched['dma_stride']
            assert sched['x_bar'] < slot - 1 and slot < 64 + 56, (slot, sched)
            if not no_dma:
                if setup:
                    put(slot - 1, setup)
                put(slot, g)
    if kind != 'last':
        y = 64 + sched['y_wait']
        put(y, 'Y_WAIT')
        put(y + 1, 's_barrier')
        for n, ins in enumerate(reads_into(0, p ^ 1)):
            if not no_reads:
                put(64 + sched['rd1_first'] + n * sched['rd1_stride'], ins)
        put(64 + 61, 's_waitcnt lgkmcnt(0)')
    # timing ablations 6 / 7 (diagnostic builds): what the batch scan's top-2 fold (knn_scan_fold.h: per score a key = score bits |
    # row index, m2 = med3(m1, m2, key), m1 = max(m1, key)) would cost in THIS four-wave form, where the 256 accumulators sit in
    # AGPRs and a wave has no partner on its SIMD to hide vector work behind.  A row tile of the scan is 8 K-tiles; one loop
    # iteration here is 2, so every iteration carries a QUARTER of a row tile's fold: 16 accumulator tiles x 4 scores x
    # (v_accvgpr_read + 3) = 256 vector instructions.  fold=1: as a burst right behind those accumulators' last MFMAs (tile p = 1,
    # k-half 1: what the live range of an accumulator allows - it is overwritten 64 MFMAs later); fold=2: one instruction behind
    # every MFMA of the iteration (the unreachable best case: as if the values could wait in spare registers).
    fold = sched.get('fold', 0)
    if fold and kind == 'steady':
        def score_ops(r, n):
            t, m1, m2 = 100 + (n & 3), 104 + 2 * ((n >> 2) & 3), 105 + 2 * ((n >> 2) & 3)
            return [f'v_accvgpr_read_b32 v{t}, a{r}', f'v_and_or_b32 v{t}, v{t}, v112, v113',
                    f'v_med3_f32 v{m2}, v{m1}, v{m2}, v{t}', f'v_max_f32 v{m1}, v{m1}, v{t}']
        if fold == 1 and p == 1:
            for at in range(16):                                   # accumulator tile `at`: last MFMA at slot 64 + at
                ops = [o for e in range(4) for o in score_ops(4 * at + e, 4 * at + e)]
                for n, o in enumerate(ops):
                    put(min(127, 64 + at + 2 + n // 8), o)
        if fold == 2:
            ops = [o for at in range(8) for e in range(4) for o in score_ops(4 * ((at + 24 + 32 * p) % 64) + e, 4 * at + e)]
            for n, o in enumerate(ops):
                put(n, o)
    out = []
    for s in range(128):
        if not sched.get('no_mfma', 0):
            out.append(mfma(s // 64, (s % 64) // 8, s % 8))
        out += extra[s]
    # the counted wait: everything but the pieces of tile t+2 issued so far
    if 'Y_WAIT' in out:
        y = out.index('Y_WAIT')
        n_before = sum(1 for ins in out[:y] if ins.startswith('buffer_load'))
        if kind == 'steady' and no_dma:
            del out[y]
        else:
            out[y] = f's_waitcnt vmcnt({n_before})' if kind == 'steady' else 's_waitcnt vmcnt(0)'
    return out


def wave_text(sched, w):
    L = [f'VQ_A256_W{w}_%=:']
    L += ['s_cmp_eq_u32 s55, 0', f's_cbranch_scc1 VQ_A256_TAIL{w}_%=', f'VQ_A256_LOOP{w}_%=:']
    L += tile(0, 'steady', sched, w) + tile(1, 'steady', sched, w)
    L += ['s_sub_u32 s55, s55, 1', 's_cmp_lg_u32 s55, 0', f's_cbranch_scc1 VQ_A256_LOOP{w}_%=', f'VQ_A256_TAIL{w}_%=:']
    L += tile(0, 'pen', sched, w) + tile(1, 'last', sched, w)
    if w < 3:
        L += ['s_branch VQ_A256_END_%=']
    return L


def build(sched):
    L = []
    # ---- setup: operands -> fixed registers ----
    L += ['s_mov_b32 s56, %[srd_a0]', 's_mov_b32 s57, %[srd_a1]', 's_mov_b32 s58, 0x7ffffffe', 's_mov_b32 s59, 0x00020000',
          's_mov_b32 s60, %[srd_w0]', 's_mov_b32 s61, %[srd_w1]', 's_mov_b32 s62, 0x7ffffffe', 's_mov_b32 s63, 0x00020000',
          'v_mov_b32 v124, %[a_v0]', 'v_xor_b32 v125, 64, v124', 'v_mov_b32 v126, %[w_v0]', 'v_xor_b32 v127, 64, v126',
          'v_mov_b32 v116, %[rd_a]', 'v_xor_b32 v117, 64, v116', 'v_mov_b32 v118, %[rd_w]', 'v_xor_b32 v119, 64, v118',
          f'v_add_u32 v120, {BUF_BYTES}, v116', f'v_add_u32 v121, {BUF_BYTES}, v117',
          f'v_add_u32 v122, {BUF_BYTES}, v118', f'v_add_u32 v123, {BUF_BYTES}, v119',
          's_mov_b32 s54, %[m0_a]', 's_mov_b32 s55, %[trips]']
    for q in range(1, 8):
        L.append(f's_mul_i32 s{S_ROW_A(q)}, %[a_row8], {q}')
        L.append(f's_mul_i32 s{S_ROW_W(q)}, %[w_row8], {q}')
    # ---- tiles 0 and 1 are in flight (issued by the C++ prologue): zero the accumulators, wait for tile 0, first fragments ----
    for r in range(256):
        L.append(f'v_accvgpr_write_b32 a{r}, 0')
    L += ['s_waitcnt vmcnt(16)', 's_barrier']
    L += reads_into(0, 0)
    L += ['s_waitcnt lgkmcnt(0)']
    # ---- one text per wave ----
    for w in (1, 2, 3):
        L += [f's_cmp_eq_u32 %[wave], {w}', f's_cbranch_scc1 VQ_A256_W{w}_%=']
    for w in range(4):
        L += wave_text(sched, w)
    L += ['VQ_A256_END_%=:', 's_nop 15', 's_nop 15']      # the last MFMAs retire before v_accvgpr_read (no hazard handling inside inline asm)
    return L


SCHED = dict(rd0_first=0, rd0_stride=1, x_wait=22, x_bar=23, dma_first=25, dma_stride=4, y_wait=12, rd1_first=14, rd1_stride=2)

VARIANTS = [            # (macro suffix, overrides): variant 0 is the product schedule, the others are timing ablations (wrong results)
    ('0', {}),
    ('1', dict(no_dma=1)),
    ('2', dict(no_dma=1, no_reads=1)),
    ('3', dict(same_slots=1)),
    ('4', dict(no_mfma=1)),
    ('5', dict(no_mfma=1, no_reads=1)),
    ('6', dict(fold=1)),
    ('7', dict(fold=2)),
]


def write_text(f, name, lines):
    f.write(f'#define VQ_A256_LOOP_TEXT_{name}(TY) \\\n')
    for ins in lines:
        if ins.startswith(MFMA):
            f.write('    "v_mfma_f32_16x16x32_" TY "' + ins[len(MFMA):] + '\\n" \\\n')
        else:
            f.write('    "' + ins + '\\n" \\\n')
    f.write('    ""\n\n')
    waits = sorted(set(l for l in lines if l.startswith('s_waitcnt vmcnt')))
    print(f'variant {name}: {len(lines)} instructions, {sum(1 for l in lines if l.startswith(MFMA))} MFMAs, counted waits {waits}')


def main():
    """usage: gen_gemm_asm.py [key=value ...] [--diag PATH]
    Writes csrc/gemm_asm256_loop.inc (the product schedule, committed); with --diag also PATH = the timing ablations
    (variants 1-5, wrong results by construction) that `make DIAG=1` builds into the diagnostic library."""
    args = sys.argv[1:]
    diag_path = None
    if '--diag' in args:
        i = args.index('--diag')
        diag_path = args[i + 1]
        del args[i:i + 2]
    base = dict(SCHED)
    for a in args:
        k, v = a.split('=')
        base[k] = int(v)
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, '..', 'video-quierer_amd', 'csrc', 'gemm_asm256_loop.inc')
    sched_note = ', '.join(f'{k}={v}' for k, v in sorted(base.items()))
    with open(path, 'w') as f:
        f.write(f'// GENERATED by scripts/gen_gemm_asm.py (schedule: {sched_note}) - do not edit.\n')
        f.write('// The K loop of gemm_tn256a_kernel (gemm_asm256.h) as one inline-asm text; TY = "f16" or "bf16".\n')
        write_text(f, '0', build(base))
        regs = [f'"a{i}"' for i in range(256)] + [f'"v{i}"' for i in range(116, 256)] + [f'"s{i}"' for i in range(40, 64)]
        f.write('#define VQ_A256_CLOBBERS \\\n')
        for i in range(0, len(regs), 16):
            f.write('    ' + ', '.join(regs[i:i + 16]) + (', \\\n' if i + 16 < len(regs) else '\n'))
        f.write('\n// acc(I, J) (16 x 16 tile: rows 16 I.., columns 16 J.. of the wave tile) -> an f32x4; I and J must be constants after unrolling.\n')
        f.write('// "memory": a read stays behind every load and store written before it, so the epilogue holds one pass of tiles at a time\n')
        f.write('#define VQ_A256_READ_TILE(I, J, DST) do { float t0_, t1_, t2_, t3_; switch ((I) * 8 + (J)) { \\\n')
        for t in range(64):
            f.write(f'    case {t}: asm volatile("v_accvgpr_read_b32 %0, a{4 * t}\\n\\tv_accvgpr_read_b32 %1, a{4 * t + 1}\\n\\tv_accvgpr_read_b32 %2, a{4 * t + 2}\\n\\tv_accvgpr_read_b32 %3, a{4 * t + 3}" : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_) : : "memory"); break; \\\n')
        f.write('    default: t0_ = t1_ = t2_ = t3_ = 0.f; } (DST) = f32x4{t0_, t1_, t2_, t3_}; } while (0)\n')
    print(path)
    if diag_path:
        with open(diag_path, 'w') as f:
            f.write(f'// GENERATED by scripts/gen_gemm_asm.py --diag (schedule: {sched_note}) - not committed.\n')
            f.write('// Timing ablations of the K loop of gemm_tn256a_kernel for the diagnostic library: their results are wrong by construction.\n')
            for name, over in VARIANTS[1:]:
                sched = dict(base); sched.update(over)
                write_text(f, name, build(sched))
            f.write('// registers the fold ablations (6, 7) use beside the product text\n')
            f.write('#define VQ_A256_DIAG_CLOBBERS , ' + ', '.join(f'"v{i}"' for i in range(100, 116)) + '\n')
        print(diag_path)


if __name__ == '__main__':
    main()
