// What this box's memory gives right now: a plain nontemporal fill and a plain read of 8 GiB, one contiguous 4 KB chunk per
// workgroup (the geometry that reaches the ceilings in profiles/r03_hbm_ceilings.txt), 7 launches each, median and spread.
// Printed beside measurements whose run-to-run / box-to-box spread is in question (profiles/r06_*): the boxes of the pool and
// the minutes on one box differ by more than most kernel changes.
//   hipcc --offload-arch=gfx950 -O3 -o box_speed tools/box_speed.hip && ./box_speed
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_fill(v4i *p)
{
    const v4i v = { 1, 2, 3, 4 };
    __builtin_nontemporal_store(v, p + (size_t) blockIdx.x * 256 + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_read(const v4i *p, int *out)
{
    const v4i v = __builtin_nontemporal_load(p + (size_t) blockIdx.x * 256 + threadIdx.x);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678) *out = 1;
}
int main()
{
    const size_t bytes = 8ull << 30, blocks = bytes / 4096;
    v4i *p; int *o;
    if (hipMalloc((void **) &p, bytes) != hipSuccess || hipMalloc((void **) &o, 4) != hipSuccess) { printf("box_speed: no memory\n"); return 1; }
    hipMemset(p, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[2][7];
    for (int k = 0; k < 2; k++) for (int r = -1; r < 7; r++) {
        hipEventRecord(a, 0);
        if (k == 0) hipLaunchKernelGGL(k_fill, dim3((unsigned) blocks), dim3(256), 0, 0, p);
        else hipLaunchKernelGGL(k_read, dim3((unsigned) blocks), dim3(256), 0, 0, p, o);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (r >= 0) ms[k][r] = t;
    }
    for (int k = 0; k < 2; k++) {
        std::sort(ms[k], ms[k] + 7);
        printf("box_speed %s 8 GiB: median %.0f GB/s (fastest %.0f, slowest %.0f)\n", k ? "read" : "fill", bytes / ms[k][3] * 1e-6, bytes / ms[k][0] * 1e-6, bytes / ms[k][6] * 1e-6);
    }
    return 0;
}
