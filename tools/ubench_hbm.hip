// HBM bandwidth ceilings for the access kinds the field-pass kernels use (MI355X): write-only streams (the decoder's
// picture), read-only streams (the encoder's image), and both at once.  hipcc --offload-arch=gfx950 -O3 -o ubench_hbm.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k_fill(v4i *p, size_t n16, int nt)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
    v4i v = { 1, 2, 3, 4 };
    for (; i < n16; i += stride) { if (nt) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
__global__ void k_read(const v4i *p, size_t n16, int *out)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
    int acc = 0;
    for (; i < n16; i += stride) { v4i v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *out = acc;
}
__global__ void k_copy(const v4i *s, v4i *d, size_t n16)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
// picture-like writes: 128-byte pieces, each block walks `rows` rows of `pitch` bytes, 4 row copies per piece (row duplication)
__global__ void k_rows(v4i *p, size_t pitch16, int rows_per_block, int dups)
{
    const int piece = threadIdx.x & 7, r = threadIdx.x >> 3;       // 8 lanes x 16 B = 128 B of one row, 32 rows per block pass
    v4i v = { 1, 2, 3, 4 };
    for (size_t x = 0; x + 8 <= pitch16; x += 8)
        for (int rr = r; rr < rows_per_block; rr += blockDim.x / 8)
            for (int d = 0; d < dups; d++)
                __builtin_nontemporal_store(v, p + ((size_t) blockIdx.x * rows_per_block * dups + (size_t) rr * dups + d) * pitch16 + x + piece);
}
int main()
{
    const size_t bytes = 12ull << 30, n16 = bytes / 16;
    v4i *a, *b; int *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double gb, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int it = 0; it < 5; it++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %8.1f GB/s  (%.3f ms for %.1f GB)\n", name, gb / (best * 1e-3), best, gb);
    };
    const int grid = 256 * 32;
    run("fill, plain 16-byte stores", bytes / 1e9, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, a, n16, 0); });
    run("fill, nontemporal 16-byte stores", bytes / 1e9, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, a, n16, 1); });
    run("read, nontemporal 16-byte loads", bytes / 1e9, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n16, o); });
    run("copy (read + write bytes counted)", 2.0 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n16); });
    // 1080p pictures: pitch 7680 B, 240 lines x 4 duplicated rows: one block = 60 lines
    const size_t pitch16 = 7680 / 16;
    const int pics = (int) (bytes / (7680ull * 1080));
    run("picture rows: 128-byte pieces x 4 duplicated rows", pics * 7680.0 * 960 / 1e9,
        [&] { hipLaunchKernelGGL(k_rows, dim3(pics * 4), dim3(256), 0, 0, a, pitch16, 60, 4); });
    return 0;
}
