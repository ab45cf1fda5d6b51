// Does global_load_lds_dwordx4 need a 16-byte aligned global address?  One wavefront copies 1 KiB from src + shift bytes (shift = 0, 4, 8, 12)
// into LDS by LDS-DMA and writes it back out; the host compares.   hipcc --offload-arch=gfx950 -O2 tools/lab/glds_align.hip -o /tmp/glds_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__global__ void k(const unsigned char *src, int shift, unsigned *out) {
    __shared__ __align__(16) unsigned buf[256];
    const unsigned l0 = (unsigned)(size_t)buf;   // LDS aperture: low 32 bits are the LDS byte address
    glds16(src + shift, threadIdx.x * 16u, l0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}
int main() {
    std::vector<unsigned char> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)(i * 7 + 3);
    unsigned char *d; unsigned *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift : {0, 4, 8, 12, 16, 24}) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift, o);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned char> r(1024);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += r[i] != h[i + shift];
        printf("shift %2d: %s, %d of 1024 bytes differ\n", shift, hipGetErrorString(e), bad);
    }
    return 0;
}
