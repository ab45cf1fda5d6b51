// Issue-rate probe for gfx950: cycles per wave64 instruction of v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32 and v_fma_f64,
// one wavefront per SIMD and four, 16 independent accumulators each (no dependent stalls).   hipcc --offload-arch=gfx950 -O3 tools/lab/pk_rate.cpp -o tools/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND> __global__ void k(long long *out, float seed) {
    f2 a[16];
    double d[16];
    for (int i = 0; i < 16; ++i) { a[i] = f2{seed + i, seed - i}; d[i] = seed + i; }
    const f2 m = f2{1.0001f, 0.9999f}, c = f2{1e-7f, -1e-7f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
#define P_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define P_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define P_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define D_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)1.0001), "v"((double)1e-7));
        if (KIND == 0) { REP16(S_FMA) }
        if (KIND == 1) { REP16(P_FMA) }
        if (KIND == 2) { REP16(P_MUL) }
        if (KIND == 3) { REP16(P_ADD) }
        if (KIND == 4) { REP16(D_FMA) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y + (float)d[i];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (long long)s; }
}
int main() {
    long long *o; hipMalloc(&o, 1 << 22);
    static long long h[1 << 16];
    const char *nm[5] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f64"};
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs; per configuration: s_memtime ticks of one wavefront for its 4096 instructions, and the whole launch's wall time\n", cus);
    // waves per SIMD: 1 block of 256 threads per CU = 1; 1024 threads = 4; two / four such blocks per CU = 8 / 16 (the grid is a multiple of the CU count)
    const int cfg[4][2] = {{256, 1}, {1024, 1}, {1024, 2}, {1024, 4}};
    for (int ci = 0; ci < 4; ++ci) {
        const int threads = cfg[ci][0], per_cu = cfg[ci][1], grid = cus * per_cu;
        for (int kind = 0; kind < 5; ++kind) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(threads), 0, 0, o, 1.0f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(threads), 0, 0, o, 1.0f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(threads), 0, 0, o, 1.0f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(threads), 0, 0, o, 1.0f); break;
                default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(threads), 0, 0, o, 1.0f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, o, 16 * grid, hipMemcpyDeviceToHost);
            long long mx = 0; for (int b = 0; b < grid; ++b) mx = h[2 * b] > mx ? h[2 * b] : mx;
            const double waves_per_simd = threads / 64.0 * per_cu / 4.0;
            // SIMD-time per instruction if the SIMD were the limit: wall time x clock / (instructions per SIMD)
            printf("%5.1f waves/SIMD  %-13s wave: %6lld ticks   launch: %.3f ms = %.2f ns per wave-instruction and SIMD\n", waves_per_simd, nm[kind], mx, ms,
                   ms * 1e6 / (4096.0 * waves_per_simd));
        }
    }
    return 0;
}
