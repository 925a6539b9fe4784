/* startup_probe.c -- where does a ONE-SHOT process spend its time before the first picture?  (BASELINE configs[0]: the `ntsc` CLI
 * linked against libntsccrt_hip_ntsc.so takes ~0.3 s where the reference binary takes 22 ms; VERDICT round 4, weak #7.)
 * Calls the drop-in API exactly like crt_main.c:146-283 does -- crt_init, then crt_modulate / crt_demodulate twice -- with the
 * wall clock around every call; `hipinit` as first argument initialises the HIP runtime by hand first (dlopen + hipInit +
 * hipGetDeviceCount), which separates the runtime's own start-up from what the library adds (context tables, code-object
 * loads at the first launch out of each translation unit, first-touch of pinned / device buffers).
 *   gcc -O2 -I include -o lib/startup_probe tools/startup_probe.c -Llib -lntsccrt_hip_ntsc -ldl        (ntsc-crt_amd/Makefile)
 */
#include "crt_core.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double) t.tv_sec + 1e-6 * (double) t.tv_nsec;
}

int main(int argc, char **argv)
{
    static struct CRT crt;
    struct NTSC_SETTINGS s;
    const int w = 640, h = 480;
    unsigned char *img = malloc((size_t) w * h * 3), *out = calloc((size_t) w * h, 4);
    double t0 = now_ms(), t1;
    int k;
    for (k = 0; k < w * h * 3; k++) img[k] = (unsigned char) (k * 2654435761u >> 24);
    if (argc > 1 && !strcmp(argv[1], "hipinit")) {
        void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        int (*init)(unsigned) = hip ? (int (*)(unsigned)) dlsym(hip, "hipInit") : 0;
        int (*count)(int *) = hip ? (int (*)(int *)) dlsym(hip, "hipGetDeviceCount") : 0;
        int (*setdev)(int) = hip ? (int (*)(int)) dlsym(hip, "hipSetDevice") : 0;
        int (*mal)(void **, size_t) = hip ? (int (*)(void **, size_t)) dlsym(hip, "hipMalloc") : 0;
        int n = 0; void *p = 0;
        t1 = now_ms(); printf("dlopen libamdhip64            %8.2f ms\n", t1 - t0); t0 = t1;
        if (init) init(0);
        t1 = now_ms(); printf("hipInit                       %8.2f ms\n", t1 - t0); t0 = t1;
        if (count) count(&n);
        if (setdev) setdev(0);
        t1 = now_ms(); printf("hipGetDeviceCount+SetDevice   %8.2f ms (%d devices)\n", t1 - t0, n); t0 = t1;
        if (mal) mal(&p, 1 << 20);
        t1 = now_ms(); printf("first hipMalloc (context)     %8.2f ms\n", t1 - t0); t0 = t1;
    }
    crt_init(&crt, w, h, CRT_PIX_FORMAT_BGRA, out);
    t1 = now_ms(); printf("crt_init                      %8.2f ms\n", t1 - t0); t0 = t1;
    memset(&s, 0, sizeof(s));
    s.data = img; s.format = CRT_PIX_FORMAT_RGB; s.w = w; s.h = h; s.as_color = 1;
    crt.blend = 1; crt.scanlines = 0;
    for (k = 0; k < 3; k++) {
        crt_modulate(&crt, &s);
        t1 = now_ms(); printf("crt_modulate   #%d             %8.2f ms\n", k + 1, t1 - t0); t0 = t1;
        crt_demodulate(&crt, 0);
        t1 = now_ms(); printf("crt_demodulate #%d             %8.2f ms\n", k + 1, t1 - t0); t0 = t1;
        s.field ^= 1;
    }
    printf("checksum %u\n", (unsigned) out[12345] + 256u * out[54321]);
    return 0;
}
