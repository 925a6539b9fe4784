// HBM bandwidth ceilings for the access kinds the field-pass kernels use (MI355X): write-only streams (the decoder's
// picture), read-only streams (the encoder's image), and both at once -- swept over launch geometry, because the
// round-2 version of this file (grid-stride loops, 8192 blocks) measured a 5.9 TB/s copy where
// /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_hbm.bin tools/ubench_hbm.hip
//   ./ubench_hbm.bin            sweep, prints the best geometry per access kind
//   ./ubench_hbm.bin calib      one launch of every calibration pattern (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE:
//                               the byte count of every kernel is in its name, tools/calib_pmc.py divides)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));

// MODE 0: grid-stride, 1: one contiguous chunk per block (U pieces per thread in flight)
template <int NT, int U>
__global__ void k_fill(v4i *p, size_t n16, int chunked)
{
    v4i v = { 1, 2, 3, 4 };
    if (chunked) {
        size_t base = (size_t) blockIdx.x * blockDim.x * U + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + (size_t) u * blockDim.x;
            if (i < n16) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
        }
    } else {
        size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
        for (; i < n16; i += stride) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
    }
}
template <int NT, int U>
__global__ void k_read(const v4i *p, size_t n16, int *out, int chunked)
{
    int acc = 0;
    if (chunked) {
        size_t base = (size_t) blockIdx.x * blockDim.x * U + threadIdx.x;
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + (size_t) u * blockDim.x;
            if (i >= n16) i = n16 - 1;
            v[u] = NT ? __builtin_nontemporal_load(p + i) : p[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    } else {
        size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
        for (; i < n16; i += stride) { v4i v = NT ? __builtin_nontemporal_load(p + i) : p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678) *out = acc;
}
template <int NT, int U>
__global__ void k_copy(const v4i *s, v4i *d, size_t n16, int chunked)
{
    if (chunked) {
        size_t base = (size_t) blockIdx.x * blockDim.x * U + threadIdx.x;
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + (size_t) u * blockDim.x;
            if (i >= n16) i = n16 - 1;
            v[u] = NT ? __builtin_nontemporal_load(s + i) : s[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + (size_t) u * blockDim.x;
            if (i < n16) { if (NT) __builtin_nontemporal_store(v[u], d + i); else d[i] = v[u]; }
        }
    } else {
        size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
        for (; i < n16; i += stride) {
            v4i v = NT ? __builtin_nontemporal_load(s + i) : s[i];
            if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v;
        }
    }
}
// picture-like writes (the decoder's store pattern): a 64-lane wave owns 64 picture LINES; per tile of PIECES*16 bytes
// it stores, for each of its lines, `dups` duplicated rows -- PIECES lanes x 16 B per row piece, 64/PIECES lines per
// instruction.  pitch16 = row pitch in 16-byte units.
template <int PIECES>
__global__ void __launch_bounds__(64) k_rows(v4i *p, size_t pitch16, int lines_per_pic, int dups, int rows_per_pic, int n_pics)
{
    const int lane = threadIdx.x, piece = lane % PIECES, lr = lane / PIECES;
    const size_t wave = blockIdx.x;
    v4i v = { 1, 2, 3, 4 };
    for (size_t x = 0; x + PIECES <= pitch16; x += PIECES) {
#pragma unroll 2
        for (int i = 0; i < PIECES; i++) {
            const size_t line = wave * 64 + (size_t) i * (64 / PIECES) + lr;
            const size_t pic = line / lines_per_pic, l = line % lines_per_pic;
            if (pic < (size_t) n_pics) {
                v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + x + piece;
                for (int d = 0; d < dups; d++) __builtin_nontemporal_store(v, row + (size_t) d * pitch16);
            }
        }
    }
}
// calibration patterns for the TCC counters (known byte counts): 64-byte row pieces at a stride, scattered 16-byte windows
__global__ void k_read_pieces64(const v4i *p, size_t n_pieces, size_t stride16, int *out)
{
    // 4 lanes x 16 B = one 64-byte piece; consecutive pieces `stride16` apart (a different 128-byte line each)
    size_t t = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t piece = t >> 2;
    int acc = 0;
    if (piece < n_pieces) { v4i v = p[piece * stride16 + (t & 3)]; acc = v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *out = acc;
}
__global__ void k_read_scatter16(const v4i *p, size_t n, size_t stride16, int *out)
{
    size_t t = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    int acc = 0;
    if (t < n) { v4i v = p[t * stride16]; acc = v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *out = acc;
}
__global__ void k_write_pieces64(v4i *p, size_t n_pieces, size_t stride16)
{
    size_t t = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t piece = t >> 2;
    v4i v = { 1, 2, 3, 4 };
    if (piece < n_pieces) p[piece * stride16 + (t & 3)] = v;
}
// 64-byte pieces at an odd byte offset (the encoder's 753-byte sample rows: two partial sectors per piece)
__global__ void k_write_pieces64_unaligned(char *p, size_t n_pieces, size_t stride_bytes, int off)
{
    size_t t = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t piece = t >> 2;
    struct __attribute__((packed)) u16 { v4i v; };
    v4i v = { 1, 2, 3, 4 };
    if (piece < n_pieces) ((u16 *) (p + piece * stride_bytes + off + (t & 3) * 16))->v = v;
}


// ---- variants of the decoder's store pattern (what makes picture-row writes slower than a plain fill?) ----
// VAR 0: baseline (as k_rows<8>)   1: plain stores   2: dups in the outer loop (all pieces of row copy 0, then copy 1, ...)
// 3: XCD-aware wave -> line-group map (consecutive groups on one XCD)   4: 1 KB of ONE row per instruction (needs a 256-px tile)
// 5: 256-thread blocks = 4 neighbouring line groups per CU slot
template <int VAR>
__global__ void k_rows_var(v4i *p, size_t pitch16, int lines_per_pic, int dups, int rows_per_pic, int n_pics, int n_waves)
{
    const int lane = threadIdx.x & 63;
    size_t wave = (size_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (VAR == 3) { const size_t per = n_waves / 8; const size_t b = blockIdx.x; wave = (b % 8) * per + b / 8; if (b >= per * 8) wave = b; }
    v4i v = { 1, 2, 3, 4 };
    if (VAR == 4) {
        for (size_t x = 0; x + 64 <= pitch16; x += 64)
            for (int i = 0; i < 64; i++) {
                const size_t line = wave * 64 + i, pic = line / lines_per_pic, l = line % lines_per_pic;
                if (pic < (size_t) n_pics) {
                    v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + x + lane;
                    for (int d = 0; d < dups; d++) __builtin_nontemporal_store(v, row + (size_t) d * pitch16);
                }
            }
        return;
    }
    const int piece = lane % 8, lr = lane / 8;
    for (size_t x = 0; x + 8 <= pitch16; x += 8) {
        if (VAR == 2) {
            for (int d = 0; d < dups; d++)
                for (int i = 0; i < 8; i++) {
                    const size_t line = wave * 64 + (size_t) i * 8 + lr, pic = line / lines_per_pic, l = line % lines_per_pic;
                    if (pic < (size_t) n_pics)
                        __builtin_nontemporal_store(v, p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic + d) * pitch16 + x + piece);
                }
        } else {
#pragma unroll 2
            for (int i = 0; i < 8; i++) {
                const size_t line = wave * 64 + (size_t) i * 8 + lr, pic = line / lines_per_pic, l = line % lines_per_pic;
                if (pic < (size_t) n_pics) {
                    v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + x + piece;
                    for (int d = 0; d < dups; d++) { if (VAR == 1) row[(size_t) d * pitch16] = v; else __builtin_nontemporal_store(v, row + (size_t) d * pitch16); }
                }
            }
        }
    }
}
// the encoder's read pattern at 1080p: a wave owns 64 image rows `row_step` rows apart, reads 128-byte pieces (8 lanes x 16 B) of
// each, DEPTH tiles in flight
template <int DEPTH, int NT>
__global__ void __launch_bounds__(64) k_rows_read(const v4i *p, size_t pitch16, int rows_per_pic, int lines_per_pic, int n_pics, int *out)
{
    const int lane = threadIdx.x, piece = lane % 8, lr = lane / 8;
    const size_t wave = blockIdx.x;
    const v4i *rowp[8];
    for (int i = 0; i < 8; i++) {
        size_t line = wave * 64 + (size_t) i * 8 + lr, pic = line / lines_per_pic, l = line % lines_per_pic;
        if (pic >= (size_t) n_pics) { pic = 0; l = 0; }
        rowp[i] = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + piece;
    }
    int acc = 0;
    v4i buf[DEPTH][8];
    const size_t tiles = pitch16 / 8;
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int i = 0; i < 8; i++) buf[d][i] = NT ? __builtin_nontemporal_load(rowp[i] + d * 8) : rowp[i][d * 8];
    for (size_t t = 0; t < tiles; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int i = 0; i < 8; i++) { acc += buf[d][i].x ^ buf[d][i].y ^ buf[d][i].z ^ buf[d][i].w; }
            const size_t tn = t + d + DEPTH;
            if (tn < tiles) {
#pragma unroll
                for (int i = 0; i < 8; i++) buf[d][i] = NT ? __builtin_nontemporal_load(rowp[i] + tn * 8) : rowp[i][tn * 8];
            }
            // stand-in for the per-tile arithmetic of k_active (keeps the loads from being issued back to back)
            for (int k = 0; k < 200; k++) acc = acc * 1664525 + 1013904223;
        }
    }
    if (acc == 0x12345678) *out = acc;
}

// round 4: store patterns of a decoder whose wave owns FEWER lines and writes WIDER runs (lane = pixel group in the pixel stage):
//   LPW lines per wave, RUN16 16-byte pieces per contiguous run of one row (64 = 1 KB, 32 = 512 B ...), dups alternate 3 / 4
//   (scanlines 1 at 1080 rows: 3.5 rows per line on average).  FULLROW: the run loop inside the line (whole rows at once).
template <int LPW, int RUN16, bool FULLROW>
__global__ void __launch_bounds__(64) k_rows_wide(v4i *p, size_t pitch16, int lines_per_pic, int rows_per_pic, int n_pics)
{
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x;
    constexpr int ROWS_PER_INSTR = 64 / RUN16;                // rows (lines) covered by one store instruction
    const int sub = lane / RUN16, piece = lane % RUN16;
    v4i v = { 1, 2, 3, 4 };
    if (FULLROW) {
        for (int i = 0; i < LPW; i += ROWS_PER_INSTR) {
            const size_t line = wave * LPW + i + sub, pic = line / lines_per_pic, l = line % lines_per_pic;
            if (pic >= (size_t) n_pics) continue;
            const int dups = 3 + (int) (l & 1);
            v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + piece;
            for (int d = 0; d < dups; d++)
                for (size_t x = 0; x + RUN16 <= pitch16; x += RUN16) __builtin_nontemporal_store(v, row + (size_t) d * pitch16 + x);
        }
        return;
    }
    for (size_t x = 0; x + RUN16 <= pitch16; x += RUN16)
        for (int i = 0; i < LPW; i += ROWS_PER_INSTR) {
            const size_t line = wave * LPW + i + sub, pic = line / lines_per_pic, l = line % lines_per_pic;
            if (pic >= (size_t) n_pics) continue;
            const int dups = 3 + (int) (l & 1);
            v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + x + piece;
            for (int d = 0; d < dups; d++) __builtin_nontemporal_store(v, row + (size_t) d * pitch16);
        }
}

// round 4: MIMIC of a decoder wave -- the store pattern of k_rows_wide with `valu` dependent integer instructions (three chains of
// sub / add / 24-bit multiply-add, the decoder's mix) in front of every tile's stores, at the occupancy the real kernel would have
// (dynamic LDS sized by the caller).  Question: does a wave that owns 16 lines, writes 1 KB runs and spends 35 % more vector
// instructions per line beat one that owns 64 lines and writes 128-byte runs -- with the vector work in the picture?
template <int LPW, int RUN16>
__global__ void __launch_bounds__(64) k_mimic(v4i *p, size_t pitch16, int lines_per_pic, int rows_per_pic, int n_pics, int valu, int *sink)
{
    extern __shared__ int s_pad[];
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x;
    constexpr int ROWS_PER_INSTR = 64 / RUN16;
    const int sub = lane / RUN16, piece = lane % RUN16;
    int a = lane, b = lane * 3 + 1, c = lane ^ 5, k1 = 40503, k2 = 7;
    for (size_t x = 0; x + RUN16 <= pitch16; x += RUN16) {
        for (int i = 0; i < valu; i += 9) {
            asm volatile("v_sub_u32 %0, %1, %0\n\tv_add_u32 %0, %0, %2\n\tv_mad_i32_i24 %0, %0, %3, %1\n\t"
                         "v_sub_u32 %4, %1, %4\n\tv_add_u32 %4, %4, %2\n\tv_mad_i32_i24 %4, %4, %3, %1\n\t"
                         "v_sub_u32 %5, %1, %5\n\tv_add_u32 %5, %5, %2\n\tv_mad_i32_i24 %5, %5, %3, %1"
                         : "+v"(a), "+v"(k2), "+v"(k1), "+v"(k1), "+v"(b), "+v"(c));
        }
        v4i v = { a, b, c, 4 };
        for (int i = 0; i < LPW; i += ROWS_PER_INSTR) {
            const size_t line = wave * LPW + i + sub, pic = line / lines_per_pic, l = line % lines_per_pic;
            if (pic >= (size_t) n_pics) continue;
            const int dups = 3 + (int) (l & 1);
            v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + x + piece;
            for (int d = 0; d < dups; d++) __builtin_nontemporal_store(v, row + (size_t) d * pitch16);
        }
    }
    if (a == 0x7fffffff) { s_pad[lane] = a; *sink = s_pad[63 - lane]; }
}

// round 4, second question: WHERE in a tile's work its stores are issued.  The wide-run decoder as built runs its filter stage (55 % of a
// tile's vector work) without a store and then alternates a scanline's pixel arithmetic with that scanline's 3-4 row stores (MODE 1);
// MODE 2 spreads the same stores evenly over ALL of the tile's vector work (what deferring the duplicated rows into the next run's
// filter stage would approach); MODE 0 = k_mimic's "all work, then all stores".
template <int MODE>
__global__ void __launch_bounds__(64) k_mimic_spread(v4i *p, size_t pitch16, int lines_per_pic, int rows_per_pic, int n_pics, int valu, int *sink)
{
    extern __shared__ int s_pad[];
    constexpr int LPW = 16, RUN16 = 64;
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x;
    int a = lane, b = lane * 3 + 1, c = lane ^ 5, k1 = 40503, k2 = 7;
    auto work = [&](int n) {
        for (int i = 0; i < n; i += 9) {
            asm volatile("v_sub_u32 %0, %1, %0\n\tv_add_u32 %0, %0, %2\n\tv_mad_i32_i24 %0, %0, %3, %1\n\t"
                         "v_sub_u32 %4, %1, %4\n\tv_add_u32 %4, %4, %2\n\tv_mad_i32_i24 %4, %4, %3, %1\n\t"
                         "v_sub_u32 %5, %1, %5\n\tv_add_u32 %5, %5, %2\n\tv_mad_i32_i24 %5, %5, %3, %1"
                         : "+v"(a), "+v"(k2), "+v"(k1), "+v"(k1), "+v"(b), "+v"(c));
        }
    };
    for (size_t x = 0; x + RUN16 <= pitch16; x += RUN16) {
        if (MODE == 0) work(valu);
        if (MODE == 1) work(valu * 55 / 100);
        for (int i = 0; i < LPW; i++) {
            if (MODE == 1) work(valu * 45 / 100 / LPW);
            if (MODE == 2) work(valu / LPW);
            const size_t line = wave * LPW + i, pic = line / lines_per_pic, l = line % lines_per_pic;
            if (pic >= (size_t) n_pics) continue;
            const int dups = 3 + (int) (l & 1);
            v4i v = { a, b, c, 4 };
            v4i *row = p + (pic * rows_per_pic + l * rows_per_pic / lines_per_pic) * pitch16 + x + lane;
            for (int d = 0; d < dups; d++) __builtin_nontemporal_store(v, row + (size_t) d * pitch16);
        }
    }
    if (a == 0x7fffffff) { s_pad[lane] = a; *sink = s_pad[63 - lane]; }
}

static hipEvent_t e0, e1;
template <class F> static double best_ms(F launch, int iters = 5)
{
    launch(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < iters; it++) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const size_t bytes = 12ull << 30, n16 = bytes / 16;
    v4i *a, *b; int *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 1 && !strcmp(argv[1], "calib")) {
        // one launch each, byte counts printed; kernel names are unique so the PMC csv can be joined by name
        const size_t n1 = 1ull << 30;   // 1 GiB worth of useful bytes per pattern
        hipLaunchKernelGGL((k_read<1, 4>), dim3(n1 / 16 / 1024), dim3(256), 0, 0, a, n1 / 16, o, 1);
        hipLaunchKernelGGL((k_read<0, 4>), dim3(n1 / 16 / 1024), dim3(256), 0, 0, a, n1 / 16, o, 1);
        hipLaunchKernelGGL((k_fill<1, 4>), dim3(n1 / 16 / 1024), dim3(256), 0, 0, b, n1 / 16, 1);
        hipLaunchKernelGGL((k_fill<0, 4>), dim3(n1 / 16 / 1024), dim3(256), 0, 0, b, n1 / 16, 1);
        // 64-byte pieces, one per 256 bytes (stride 16 x 16 B): useful bytes n1, 4 GiB span
        hipLaunchKernelGGL(k_read_pieces64, dim3(n1 / 64 * 4 / 256), dim3(256), 0, 0, a, n1 / 64, (size_t) 16, o);
        hipLaunchKernelGGL(k_write_pieces64, dim3(n1 / 64 * 4 / 256), dim3(256), 0, 0, b, n1 / 64, (size_t) 16);
        hipLaunchKernelGGL(k_write_pieces64_unaligned, dim3(n1 / 64 * 4 / 256), dim3(256), 0, 0, (char *) b, n1 / 64, (size_t) 256, 28);
        // scattered 16-byte windows, one per 256 bytes: useful bytes n1 / 4
        hipLaunchKernelGGL(k_read_scatter16, dim3(n1 / 64 / 256), dim3(256), 0, 0, a, n1 / 64, (size_t) 16, o);
        hipDeviceSynchronize();
        printf("calib: k_read<1,4> %zu B nt 16B/lane | k_read<0,4> %zu B plain | k_fill<1,4> %zu B nt | k_fill<0,4> %zu B plain | "
               "k_read_pieces64 %zu B | k_write_pieces64 %zu B | k_write_pieces64_unaligned %zu B | k_read_scatter16 %zu B\n",
               n1, n1, n1, n1, n1, n1, n1, n1 / 4);
        return 0;
    }
    printf("# tools/ubench_hbm.hip on MI355X, %.1f GB per test, best of 5; GB/s = useful bytes / time (copy: read + written)\n", bytes / 1e9);
    struct Best { double gbps = 0; char what[96] = ""; } bf, br, bc;
    auto note = [&](Best &bst, const char *kind, const char *geom, double gb, double ms) {
        const double g = gb / (ms * 1e-3);
        printf("%-6s %-44s %8.1f GB/s (%.3f ms)\n", kind, geom, g, ms);
        if (g > bst.gbps) { bst.gbps = g; snprintf(bst.what, sizeof(bst.what), "%s", geom); }
    };
    char geom[96];
    const int blocks[] = { 256, 512, 1024 };
    const bool rows_only = argc > 1 && !strcmp(argv[1], "rows");      // only the picture-row patterns
    for (int bi = 0; bi < (rows_only ? 0 : 3); bi++) {
        const int bs = blocks[bi];
        for (int mult = 2; mult <= 64; mult *= 2) {          // grid-stride: 256 CUs x mult blocks
            const int grid = 256 * mult;
            snprintf(geom, sizeof(geom), "grid-stride nt  block %4d grid %6d", bs, grid);
            note(bf, "fill", geom, bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_fill<1, 1>), dim3(grid), dim3(bs), 0, 0, a, n16, 0); }));
            note(br, "read", geom, bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_read<1, 1>), dim3(grid), dim3(bs), 0, 0, a, n16, o, 0); }));
            note(bc, "copy", geom, 2.0 * bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_copy<1, 1>), dim3(grid), dim3(bs), 0, 0, a, b, n16, 0); }));
        }
#define CHUNKED(U) do { \
            const unsigned grid = (unsigned) ((n16 + (size_t) bs * U - 1) / ((size_t) bs * U)); \
            snprintf(geom, sizeof(geom), "chunked nt      block %4d x%d (%u blocks)", bs, U, grid); \
            note(bf, "fill", geom, bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_fill<1, U>), dim3(grid), dim3(bs), 0, 0, a, n16, 1); })); \
            note(br, "read", geom, bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_read<1, U>), dim3(grid), dim3(bs), 0, 0, a, n16, o, 1); })); \
            note(bc, "copy", geom, 2.0 * bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_copy<1, U>), dim3(grid), dim3(bs), 0, 0, a, b, n16, 1); })); \
            snprintf(geom, sizeof(geom), "chunked plain   block %4d x%d (%u blocks)", bs, U, grid); \
            note(bf, "fill", geom, bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_fill<0, U>), dim3(grid), dim3(bs), 0, 0, a, n16, 1); })); \
            note(br, "read", geom, bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_read<0, U>), dim3(grid), dim3(bs), 0, 0, a, n16, o, 1); })); \
            note(bc, "copy", geom, 2.0 * bytes / 1e9, best_ms([&] { hipLaunchKernelGGL((k_copy<0, U>), dim3(grid), dim3(bs), 0, 0, a, b, n16, 1); })); \
        } while (0)
        CHUNKED(1); CHUNKED(2); CHUNKED(4); CHUNKED(8);
    }
    printf("BEST fill %8.1f GB/s  [%s]\nBEST read %8.1f GB/s  [%s]\nBEST copy %8.1f GB/s  [%s]\n", bf.gbps, bf.what, br.gbps, br.what, bc.gbps, bc.what);
    // the decoder's store pattern at 1080p: pitch 7680 B, 240 lines per picture, each written to 3 or 4 rows of 1080
    const size_t pitch16 = 7680 / 16;
    const int pics = (int) (bytes / (7680ull * 1080));
    if (argc > 1 && !strcmp(argv[1], "spread")) {                     // only: where in a tile's work the stores are issued
        const int mp = pics < 1536 ? pics : 1536;
        const double gbm = (double) mp * 240 * 3.5 * 7680.0 / 1e9;
#define SPREAD(MODE, VALU, LDSB, name) do { const double ms = best_ms([&] { \
            hipLaunchKernelGGL((k_mimic_spread<MODE>), dim3((mp * 240 + 15) / 16), dim3(64), LDSB, 0, a, pitch16, 240, 1080, mp, VALU, o); }, 3); \
            const double frac = (double) (pitch16 / 64 * 64) / (double) pitch16;      /* 7 whole runs of the row's 7.5 */ \
            printf("%-6s %-70s %8.3f ms  %8.1f GB/s\n", "spread", name, ms, gbm * frac / (ms * 1e-3)); } while (0)
            SPREAD(0, 0, 13472, "16 lines, 1 KB runs, stores only, 12 waves/CU");
            SPREAD(0, 4200, 13472, "4200 VALU/tile: all work, then all stores of the tile");
            SPREAD(1, 4200, 13472, "4200 VALU/tile: 55 % without stores, then work / stores per scanline  [= k_decode_wide]");
            SPREAD(2, 4200, 13472, "4200 VALU/tile: stores spread evenly over all the work");
            SPREAD(1, 3400, 13472, "3400 VALU/tile, as k_decode_wide");
            SPREAD(2, 3400, 13472, "3400 VALU/tile, stores spread evenly");
        return 0;
    }
    const int waves = (pics * 240 + 63) / 64;
    for (int dups = 1; dups <= 4; dups++) {
        const double gb = (double) pics * 240 * dups * 7680.0 / 1e9;
        snprintf(geom, sizeof(geom), "picture rows, 128-byte pieces x %d rows", dups);
        printf("%-6s %-44s %8.1f GB/s\n", "rows", geom, gb / (best_ms([&] { hipLaunchKernelGGL((k_rows<8>), dim3(waves), dim3(64), 0, 0, a, pitch16, 240, dups, 1080, pics); }) * 1e-3));
        snprintf(geom, sizeof(geom), "picture rows, 256-byte pieces x %d rows", dups);
        printf("%-6s %-44s %8.1f GB/s\n", "rows", geom, gb / (best_ms([&] { hipLaunchKernelGGL((k_rows<16>), dim3(waves), dim3(64), 0, 0, a, pitch16, 240, dups, 1080, pics); }) * 1e-3));
        snprintf(geom, sizeof(geom), "picture rows, 64-byte pieces x %d rows", dups);
        printf("%-6s %-44s %8.1f GB/s\n", "rows", geom, gb / (best_ms([&] { hipLaunchKernelGGL((k_rows<4>), dim3(waves), dim3(64), 0, 0, a, pitch16, 240, dups, 1080, pics); }) * 1e-3));
    }
    {
        const int dups = 4;
        const double gb = (double) pics * 240 * dups * 7680.0 / 1e9;
#define VARIANT(V, name, blk) do { const int wpb = (blk) / 64; \
            printf("%-6s %-44s %8.1f GB/s\n", "rowsv", name, gb / (best_ms([&] { hipLaunchKernelGGL((k_rows_var<V>), dim3((waves + wpb - 1) / wpb), dim3(blk), 0, 0, a, pitch16, 240, dups, 1080, pics, waves); }) * 1e-3)); } while (0)
        VARIANT(0, "baseline nt, 128 B x 4 rows", 64);
        VARIANT(1, "plain stores", 64);
        VARIANT(2, "row copies in the outer loop", 64);
        VARIANT(3, "XCD-aware line-group map", 64);
        VARIANT(4, "1 KB of one row per instruction", 64);
        VARIANT(5, "256-thread blocks", 256);
        {
            const double gbw = (double) pics * 240 * 3.5 * 7680.0 / 1e9;
#define WIDE(LPW, RUN16, FULL, name) printf("%-6s %-44s %8.1f GB/s\n", "rowsw", name, gbw / (best_ms([&] { \
            hipLaunchKernelGGL((k_rows_wide<LPW, RUN16, FULL>), dim3((pics * 240 + LPW - 1) / LPW), dim3(64), 0, 0, a, pitch16, 240, 1080, pics); }) * 1e-3))
            WIDE(64, 8, false, "64 lines/wave, 128 B runs, 3.5 rows");
            WIDE(64, 64, false, "64 lines/wave, 1 KB runs, 3.5 rows");
            WIDE(32, 32, false, "32 lines/wave, 512 B runs, 3.5 rows");
            WIDE(32, 64, false, "32 lines/wave, 1 KB runs, 3.5 rows");
            WIDE(16, 16, false, "16 lines/wave, 256 B runs, 3.5 rows");
            WIDE(16, 32, false, "16 lines/wave, 512 B runs, 3.5 rows");
            WIDE(16, 64, false, "16 lines/wave, 1 KB runs, 3.5 rows");
            WIDE(8, 64, false, "8 lines/wave, 1 KB runs, 3.5 rows");
            WIDE(16, 64, true, "16 lines/wave, whole rows at once");
            WIDE(4, 64, true, "4 lines/wave, whole rows at once");
            WIDE(1, 64, true, "1 line/wave, whole rows at once");
        }
        {
            // 1536 pictures (9.9 GB of stores; the bench batch of 2048 would be 13.2 GB)
            const int mp = pics < 1536 ? pics : 1536;          // (the 12 GiB buffer holds 1553 pictures of 1080 rows; the bench batch is 2048)
            const double gbm = (double) mp * 240 * 3.5 * 7680.0 / 1e9;
#define MIMIC(LPW, RUN16, VALU, LDSB, name) do { const double ms = best_ms([&] { \
            hipLaunchKernelGGL((k_mimic<LPW, RUN16>), dim3((mp * 240 + LPW - 1) / LPW), dim3(64), LDSB, 0, a, pitch16, 240, 1080, mp, VALU, o); }, 3); \
            /* (a row of 480 pieces is 7.5 runs of 64: the loop writes the 7 whole ones) */ \
            const double frac = (double) (pitch16 / (RUN16) * (RUN16)) / (double) pitch16; \
            printf("%-6s %-70s %8.3f ms  %8.1f GB/s\n", "mimic", name, ms, gbm * frac / (ms * 1e-3)); } while (0)
            MIMIC(64, 8, 0, 14080, "64 lines/wave, 128 B runs, stores only, 11 waves/CU");
            MIMIC(64, 8, 1670, 14080, "64 lines/wave, 128 B runs, 1670 VALU/tile (100 k/wave), 11 waves/CU  [= today]");
            MIMIC(64, 8, 1670, 9984, "64 lines/wave, 128 B runs, 1670 VALU/tile, 16 waves/CU");
            MIMIC(16, 64, 0, 17408, "16 lines/wave, 1 KB runs, stores only, 9 waves/CU");
            MIMIC(16, 64, 4500, 17408, "16 lines/wave, 1 KB runs, 4500 VALU/tile (34 k/wave = +35 % per line), 9 waves/CU  [= the wide-run decoder]");
            MIMIC(16, 64, 3330, 17408, "16 lines/wave, 1 KB runs, 3330 VALU/tile (25 k/wave = today's per line), 9 waves/CU");
            MIMIC(16, 64, 4500, 10240, "16 lines/wave, 1 KB runs, 4500 VALU/tile, 16 waves/CU");
            MIMIC(32, 64, 9000, 20480, "32 lines/wave, 1 KB runs, 9000 VALU/tile (68 k/wave), 8 waves/CU");
        }
        // dense pictures (no skipped rows: 960-row pictures)
        printf("%-6s %-44s %8.1f GB/s\n", "rowsv", "baseline, dense 960-row pictures", gb / (best_ms([&] { hipLaunchKernelGGL((k_rows_var<0>), dim3(waves), dim3(64), 0, 0, a, pitch16, 240, dups, 960, pics, waves); }) * 1e-3));
        // reads: 236 of 1080 rows per picture
        const int rwaves = (pics * 236 + 63) / 64;
        const double rgb = (double) pics * 236 * 7680.0 / 1e9;
#define RVAR(D, NT_) printf("%-6s read rows 128-byte pieces depth %d %s            %8.1f GB/s\n", "rowsr", D, NT_ ? "nt   " : "plain", \
            rgb / (best_ms([&] { hipLaunchKernelGGL((k_rows_read<D, NT_>), dim3(rwaves), dim3(64), 0, 0, a, pitch16, 1080, 236, pics, o); }) * 1e-3))
        RVAR(1, 1); RVAR(2, 1); RVAR(3, 1); RVAR(1, 0); RVAR(2, 0);
    }
    return 0;
}
