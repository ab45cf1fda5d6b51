// Accuracy probe for the hardware v_sin_f32 / v_cos_f32 (input in turns) against sincospif, on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/lab/trig_probe.cpp -o tools/trig_probe && tools/trig_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float *a, float *hs, float *hc, float *ls, float *lc, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = a[i];                       // angle in units of pi, [-1, 1]
    hs[i] = __builtin_amdgcn_sinf(0.5f * x);
    hc[i] = __builtin_amdgcn_cosf(0.5f * x);
    float s, c;
    sincospif(x, &s, &c);
    ls[i] = s; lc[i] = c;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), hs(n), hc(n), ls(n), lc(n);
    for (int i = 0; i < n; i++) a[i] = -1.0f + 2.0f * (float)i / (float)(n - 1);
    float *da, *d1, *d2, *d3, *d4;
    hipMalloc(&da, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4); hipMalloc(&d4, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, d1, d2, d3, d4, n);
    hipMemcpy(hs.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), d2, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ls.data(), d3, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lc.data(), d4, n * 4, hipMemcpyDeviceToHost);
    double eh = 0, el = 0;
    for (int i = 0; i < n; i++) {
        const double t = M_PI * (double)a[i];
        eh = fmax(eh, fmax(fabs(hs[i] - sin(t)), fabs(hc[i] - cos(t))));
        el = fmax(el, fmax(fabs(ls[i] - sin(t)), fabs(lc[i] - cos(t))));
    }
    printf("{\"max_abs_err_v_sin_cos\": %.3e, \"max_abs_err_sincospif\": %.3e}\n", eh, el);
    return 0;
}
