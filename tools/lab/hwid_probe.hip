// hwid_probe.hip -- where do the waves of a persistent 2-workgroups-per-CU launch land?  (lab aid, not part of the product)
// Build: hipcc -O2 --offload-arch=gfx950 hwid_probe.hip -o hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned *out)
{
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that all 512 are placed before the first one leaves
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 200000ull) { }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (lds[threadIdx.x] == 1.234) out[0] = 0;
}
int main()
{
    const int G = 512;
    unsigned *d;
    hipMalloc(&d, G * 4 * 2 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 78 * 1024, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 8);
    hipMemcpy(h.data(), d, G * 8 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu_blocks;
    int simd_in_order = 0, total = 0;
    for (int b = 0; b < G; b++) {
        unsigned key = 0;
        for (int w = 0; w < 4; w++) {
            const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            if (b < 4 || (b >= 256 && b < 260)) printf("block %3d wave %d: xcc %u se %u sh %u cu %2u simd %u wave_id %u\n", b, w, xcc, se, sh, cu, simd, hw & 15);
            simd_in_order += (simd == (unsigned)w);
            total++;
        }
        cu_blocks[key].push_back(b);
    }
    printf("waves whose SIMD id equals their index in the workgroup: %d of %d\n", simd_in_order, total);
    int pairs_256 = 0, cus = 0, other = 0;
    for (auto &kv : cu_blocks) {
        cus++;
        if (kv.second.size() == 2 && kv.second[1] - kv.second[0] == 256) pairs_256++;
        else { other++; if (other <= 8) { printf("cu key %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    }
    printf("CUs used: %d; CUs holding blocks (b, b+256): %d\n", cus, pairs_256);
    return 0;
}
