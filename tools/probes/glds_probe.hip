// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS or leave it untouched?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t* src, int bytes, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* s32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < 512; i += 64) s32[i] = 0xDEADBEEFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const int lane = threadIdx.x;
    // lanes 0..31 in range (permuted source), lanes 32..63 out of range
    unsigned voff = lane < 32 ? (unsigned)((lane ^ 5) * 16) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s32[i];
}

int main() {
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
    uint32_t *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, 4096, o);
    std::vector<uint32_t> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    for (int lane : {0, 1, 5, 31, 32, 33, 63}) {
        printf("lane %2d: %08x %08x %08x %08x\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
    }
    printf("expect lane0 <- chunk 5: %08x ; OOB lanes: 0 = zero fill, deadbeef = untouched\n", 0x1000u + 5 * 4);
    return 0;
}
