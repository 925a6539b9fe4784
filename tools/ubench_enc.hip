// Memory shapes of the 1920x1080 ENCODER (k_active), without its arithmetic: what does the memory system give a wave that
// owns 64 destination rows (128-byte image pieces, 64- or 256-byte signal pieces: today's kernel) against a wave that owns
// 16 rows (image pixels gathered lane-per-sample or fetched in 512-byte pieces, WHOLE 753-byte signal lines stored at the
// end)?  Geometry of BASELINE configs[2]: 240 destination rows per field, source rows 4.5 image rows apart, 7680-byte image
// rows, signal lines 910 bytes apart starting at an odd offset (crt_ntsc.c:254-324; DESIGN.md section 5.2, 9 lead 5).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_enc.bin tools/ubench_enc.hip
//   tools/ubench_enc.bin [fields]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) unaligned16 { v4i v; };
typedef __attribute__((address_space(1))) unaligned16 g_u16;
typedef v4i v4i_a4 __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) v4i_a4 g_v4i;
typedef __attribute__((address_space(1))) unsigned g_u32;
__device__ __forceinline__ void st16p(unsigned long long a, v4i v) { ((g_u16 *) a)->v = v; }
__device__ __forceinline__ void st16n(unsigned long long a, v4i v) { __builtin_nontemporal_store(v, (g_v4i *) a); }
#define st16(a, v) do { if (c_nt) st16n(a, v); else st16p(a, v); } while (0)
__device__ __forceinline__ v4i ld16nt(unsigned long long a) { return __builtin_nontemporal_load((const g_v4i *) a); }
__device__ __forceinline__ unsigned ld32nt(unsigned long long a) { return __builtin_nontemporal_load((const g_u32 *) a); }
__device__ __forceinline__ unsigned ld32(unsigned long long a) { return *(const g_u32 *) a; }
__device__ __forceinline__ v4i ld16(unsigned long long a) { return ((const g_u16 *) a)->v; }

#define ROWS 240
#ifdef GEO640                     /* 640x480: 2560-byte image rows, every second row, 753 samples from 640 pixels */
#define IPITCH 2560
#define SRC_ROW(y) ((y) * 2)
#define IMG_ROWS 481
#define PIXELS 640
#else
#define IPITCH 7680
#define SRC_ROW(y) ((y) * 9 / 2)
#define IMG_ROWS 1081
#define PIXELS 1920
#endif
static __device__ __constant__ int c_lpitch = 910, c_line0 = 21 * 910 + 152, c_nt = 0;
#define LPITCH c_lpitch
#define LINE0 c_line0
#define DESTW 753
#define XSTR2(x) #x
#define XSTR(x) XSTR2(x)

__device__ __forceinline__ int filler(int acc, int n)
{
    for (int i = 0; i < n; i++) acc = (acc << 1) ^ (acc + 0x9e37);      /* 3 dependent full-rate instructions */
    return acc;
}

// today's shape: a wave owns 64 rows; per tile 8 load instructions cover 64 rows x 128 bytes; the signal leaves in
// PIECE-byte runs per row (64: every 5.1 tiles; 256: every 20.4)
template <int PIECE, int IPIECE = 128>
__global__ void __launch_bounds__(64)
k_rows64(const unsigned char *img, size_t istride, unsigned char *dst, size_t fstride, int total, int do_load, int do_store, int valu, int *sink)
{
    extern __shared__ unsigned char pad[];
    __shared__ unsigned long long s_src[64], s_dst[64];
    const int lane = threadIdx.x, gid = blockIdx.x * 64 + lane;
    const int f = gid < total ? gid / ROWS : 0, y = gid < total ? gid % ROWS : 0;
    s_src[lane] = (unsigned long long) (img + (size_t) f * istride + (size_t) SRC_ROW(y) * IPITCH);
    s_dst[lane] = gid < total ? (unsigned long long) (dst + (size_t) f * fstride + LINE0 + y * LPITCH) : 0ull;
    __syncthreads();
    int acc = lane;
    constexpr int PPR = PIECE / 16, RPI = 64 / PPR;       // pieces per row, rows per store instruction
    int stored = 0;
    constexpr int NTILES = IPITCH / IPIECE, IPPR = IPIECE / 16, IRPI = 64 / IPPR;      // image pieces per row, rows per load instruction
    for (int tile = 0; tile < NTILES; tile++) {
        if (do_load) {
            v4i v[IPPR];
#pragma unroll
            for (int i = 0; i < IPPR; i++) {
                const unsigned long long a = s_src[i * IRPI + lane / IPPR] + tile * IPIECE + (lane % IPPR) * 16;
                v[i] = IPIECE == 128 ? ld16nt(a) : ld16(a);               // (narrow tiles: plain loads, the line's other half comes later)
            }
#pragma unroll
            for (int i = 0; i < IPPR; i++) acc ^= v[i].x + v[i].y + v[i].z + v[i].w;
        }
        acc = filler(acc, valu);
        // samples produced so far
        const int have = (tile + 1) * (IPIECE / 4) * DESTW / PIXELS;
        while (do_store && (have - stored >= PIECE || (tile == NTILES - 1 && stored + PIECE <= DESTW))) {
#pragma unroll 2
            for (int i = 0; i < 64 / RPI; i++) {
                const unsigned long long d = s_dst[i * RPI + lane / PPR];
                v4i o = { acc, acc + i, acc, acc };
                if (d) st16(d + stored + (lane % PPR) * 16, o);
            }
            stored += PIECE;
        }
    }
    if (acc == 0x12345) *sink = acc;
    if (pad[lane] == 77 && acc == 3) *sink = 1;
}

// 16 rows per wave.  LOADS 1: lane = sample, one dword gather per row and 64 samples (stride 10.2 bytes);
// 2: 512-byte pieces, 2 rows per instruction.  The 16 signal lines leave as whole lines at the end.
template <int LOADS>
__global__ void __launch_bounds__(64)
k_rows16(const unsigned char *img, size_t istride, unsigned char *dst, size_t fstride, int total, int do_load, int do_store, int valu, int *sink)
{
    extern __shared__ unsigned char pad[];
    __shared__ unsigned long long s_src[16], s_dst[16];
    const int lane = threadIdx.x;
    if (lane < 16) {
        const int gid = blockIdx.x * 16 + lane;
        const int f = gid < total ? gid / ROWS : 0, y = gid < total ? gid % ROWS : 0;
        s_src[lane] = (unsigned long long) (img + (size_t) f * istride + (size_t) SRC_ROW(y) * IPITCH);
        s_dst[lane] = gid < total ? (unsigned long long) (dst + (size_t) f * fstride + LINE0 + y * LPITCH) : 0ull;
    }
    __syncthreads();
    int acc = lane;
    if (LOADS == 1) {
        for (int t = 0; t < 12; t++) {
            int x = t * 64 + lane;
            if (x > DESTW - 1) x = DESTW - 1;
            const int col = x * PIXELS / DESTW;
            if (do_load) {
                unsigned v[16];
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = ld32nt(s_src[r] + col * 4);
#pragma unroll
                for (int r = 0; r < 16; r++) acc ^= v[r];
            }
            acc = filler(acc, valu);
        }
    } else {
        for (int t = 0; t < IPITCH / 512; t++) {
            if (do_load) {
                v4i v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = ld16nt(s_src[2 * i + (lane >> 5)] + t * 512 + (lane & 31) * 16);
#pragma unroll
                for (int i = 0; i < 8; i++) acc ^= v[i].x + v[i].y + v[i].z + v[i].w;
            }
            acc = filler(acc, valu);
        }
    }
    if (do_store) {
        for (int q = lane; q < 16 * 47; q += 64) {           // 47 x 16 = 752 bytes per line
            const int r = q / 47, p = q - r * 47;
            const unsigned long long d = s_dst[r];
            v4i o = { acc, acc + q, acc, acc };
            if (d) st16(d + p * 16, o);
        }
    }
    if (acc == 0x12345) *sink = acc;
    if (pad[lane] == 77 && acc == 3) *sink = 1;
}

static hipEvent_t e0, e1;
template <class F> static double best_ms(F launch, int iters = 4)
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
    const int fields = argc > 1 ? atoi(argv[1]) : 2048;
    const int uniq = fields;                                      // every field its own image (8.3 MB each)
    const size_t istride = (size_t) IMG_ROWS * IPITCH, fstride = 270336;
    unsigned char *img, *dst; int *sink;
    if (hipMalloc(&img, istride * uniq) != hipSuccess || hipMalloc(&dst, fstride * fields + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(img, 3, istride * uniq); hipMemset(dst, 0, fstride * fields);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int total = fields * ROWS;
    const double gb_in = (double) total * IPITCH / 1e9, gb_out = (double) total * 752 / 1e9;
    printf("# tools/ubench_enc.hip: %d fields of " XSTR(PIXELS) " pixels per row -> %d rows; image rows %.2f GB, signal %.2f GB; ms (best of 4)\n", fields, total, gb_in, gb_out);
    printf("# %-52s %9s %9s %9s\n", "shape (LDS pad -> waves per CU)", "loads", "stores", "both");
    // (the same istride for fields beyond `uniq`: f % uniq through a smaller stride would change the pattern; instead cap)
    const int f_eff = fields <= uniq ? fields : uniq;
    const int total_eff = f_eff * ROWS;
    const double scale = (double) total / total_eff;
    auto row = [&](const char *name, auto launch) {
        double t[3];
        const int modes[3][2] = { { 1, 0 }, { 0, 1 }, { 1, 1 } };
        for (int m = 0; m < 3; m++) t[m] = best_ms([&] { launch(modes[m][0], modes[m][1]); }) * scale;
        printf("  %-52s %9.3f %9.3f %9.3f\n", name, t[0], t[1], t[2]);
    };
    if (argc > 2 && !strcmp(argv[2], "calib")) {
        // one loads-only launch of the encoder's two image-read patterns (run under rocprofv3 --pmc FETCH_SIZE; tools/calib_pmc.py
        // divides the useful bytes printed here by the counter): 64-byte pieces whose other half-line follows a tile later
        // (k_active with the narrow image tile) and whole 128-byte lines (wide tile)
        hipLaunchKernelGGL((k_rows64<64, 64>), dim3((total_eff + 63) / 64), dim3(64), 9700, 0, img, istride, dst, fstride, total_eff, 1, 0, 0, sink);
        hipLaunchKernelGGL((k_rows64<64, 128>), dim3((total_eff + 63) / 64), dim3(64), 13800, 0, img, istride, dst, fstride, total_eff, 1, 0, 0, sink);
        hipDeviceSynchronize();
        printf("calib: k_rows64<64, 64> %.0f B | k_rows64<64, 128> %.0f B (image rows of %d fields, " XSTR(PIXELS) " pixels per row)\n",
               (double) total_eff * IPITCH, (double) total_eff * IPITCH, fields);
        return 0;
    }
    const int geo[3][2] = { { 910, 21 * 910 + 152 }, { 912, 21 * 912 + 160 }, { 1024, 21 * 1024 + 128 } };
    const char *geo_name[3] = { "reference pitch 910, odd start", "pitch 912, lines 16-byte aligned", "pitch 1024, lines 128-byte aligned" };
    for (int g = 0; g < 3; g++) for (int nt = 0; nt < 2; nt++) {
        hipMemcpyToSymbol(HIP_SYMBOL(c_lpitch), &geo[g][0], 4); hipMemcpyToSymbol(HIP_SYMBOL(c_line0), &geo[g][1], 4); hipMemcpyToSymbol(HIP_SYMBOL(c_nt), &nt, 4);
        printf("## %s; %s stores\n", geo_name[g], nt ? "nontemporal" : "plain");
        for (int pass = 0; pass < 2; pass++) {
            const int v64 = pass ? 565 / 3 : 0, v16p = pass ? 610 / 3 : 0;
            printf("# %s\n", pass ? "with the kernels' dependent vector work per tile (565 / 610 instructions)" : "memory only");
            char nm[96];
            snprintf(nm, sizeof nm, "64 rows/wave, 64 B image, 64 B signal pieces, LDS 9700");
            row(nm, [&](int l, int s) { hipLaunchKernelGGL((k_rows64<64, 64>), dim3((total_eff + 63) / 64), dim3(64), 9700, 0, img, istride, dst, fstride, total_eff, l, s, v64 / 2, sink); });
            snprintf(nm, sizeof nm, "64 rows/wave, 64 B image, 128 B signal pieces, LDS 13800");
            row(nm, [&](int l, int s) { hipLaunchKernelGGL((k_rows64<128, 64>), dim3((total_eff + 63) / 64), dim3(64), 13800, 0, img, istride, dst, fstride, total_eff, l, s, v64 / 2, sink); });
            snprintf(nm, sizeof nm, "64 rows/wave, 128 B image, 64 B signal pieces, LDS 13800");
            row(nm, [&](int l, int s) { hipLaunchKernelGGL((k_rows64<64>), dim3((total_eff + 63) / 64), dim3(64), 13800, 0, img, istride, dst, fstride, total_eff, l, s, v64, sink); });
            snprintf(nm, sizeof nm, "64 rows/wave, 128 B image, 128 B signal pieces, LDS 18000");
            row(nm, [&](int l, int s) { hipLaunchKernelGGL((k_rows64<128>), dim3((total_eff + 63) / 64), dim3(64), 18000, 0, img, istride, dst, fstride, total_eff, l, s, v64, sink); });
            snprintf(nm, sizeof nm, "64 rows/wave, 128 B image, 256 B signal pieces, LDS 26000");
            row(nm, [&](int l, int s) { hipLaunchKernelGGL((k_rows64<256>), dim3((total_eff + 63) / 64), dim3(64), 26000, 0, img, istride, dst, fstride, total_eff, l, s, v64, sink); });
            snprintf(nm, sizeof nm, "16 rows/wave, 512 B pieces, whole lines, LDS 12000");
            row(nm, [&](int l, int s) { hipLaunchKernelGGL((k_rows16<2>), dim3((total_eff + 15) / 16), dim3(64), 12000, 0, img, istride, dst, fstride, total_eff, l, s, v16p, sink); });
        }
    }
    return 0;
}
