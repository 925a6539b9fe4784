// tools/probe_mfma_i8.hip -- checks on the device the operand layout of v_mfma_i32_32x32x32_i8 that vhs_jump_mfma
// (ntsc-crt_amd/csrc/crt_noise.hip) assumes:  A fragment: row = lane & 31, k = 16 * (lane >> 5) + byte;  B fragment: column =
// lane & 31, the same k slots;  D: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Asymmetric random A and B, compared with the host product.  Prints which hypothesis holds.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_mfma_i8.bin tools/probe_mfma_i8.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const signed char *A /* [32][32] row-major: A[i][k] */, const signed char *B /* [32][32]: B[k][j] */, int *D /* 64 lanes x 16 regs */)
{
    const int lane = threadIdx.x, half = lane >> 5, rc = lane & 31;
    signed char a[16], b[16];
    for (int s = 0; s < 16; s++) { a[s] = A[rc * 32 + 16 * half + s]; b[s] = B[(16 * half + s) * 32 + rc]; }
    v4i av, bv;
    av.x = *(int *) &a[0]; av.y = *(int *) &a[4]; av.z = *(int *) &a[8]; av.w = *(int *) &a[12];
    bv.x = *(int *) &b[0]; bv.y = *(int *) &b[4]; bv.z = *(int *) &b[8]; bv.w = *(int *) &b[12];
    v16i acc = { 0 };
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[lane * 16 + r] = acc[r];
}

int main()
{
    signed char hA[1024], hB[1024];
    srand(7);
    for (int i = 0; i < 1024; i++) { hA[i] = (signed char) (rand() % 256 - 128); hB[i] = (signed char) (rand() % 256 - 128); }
    static int ref[32][32];
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int kk = 0; kk < 32; kk++) s += hA[i * 32 + kk] * hB[kk * 32 + j]; ref[i][j] = s; }
    signed char *dA, *dB; int *dD; int hD[1024];
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("probe: HIP error\n"); return 2; }
    int bad = 0, bad_t = 0;
    for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 16; r++) {
        const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (hD[lane * 16 + r] != ref[row][col]) bad++;
        if (hD[lane * 16 + r] != ref[col][row]) bad_t++;
    }
    printf("probe_mfma_i8: assumed layout %s (%d mismatches); transposed-output hypothesis %s (%d)\n",
           bad ? "WRONG" : "OK", bad, bad_t ? "no" : "YES", bad_t);
    return bad ? 1 : 0;
}
