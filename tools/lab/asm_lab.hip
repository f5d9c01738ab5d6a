// asm_lab.hip -- ablation harness for the assembly kernel (profiling aid, not part of the product).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../include -I../../fem-shell_amd/csrc asm_lab.hip ../../fem-shell_amd/csrc/plan.cpp -o asm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include "assemble_kernel.hpp"
using namespace femshell;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int W, int F> float run(const DeviceMatrix &m, const MatConst &mc, int grid, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // the first ~20 launches of a process run ~12 % slow (clocks, TLBs): warm up before timing
    static bool warm = false;
    for (int i = 0; i < (warm ? 2 : 25); i++) hipLaunchKernelGGL((k_assemble<W, F>), dim3(grid), dim3(256), m.lds_bytes, 0, m, mc);
    warm = true;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_assemble<W, F>), dim3(grid), dim3(256), m.lds_bytes, 0, m, mc);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <class T> T *up(const std::vector<T> &v) { T *d = nullptr; CK(hipMalloc(&d, std::max<size_t>(1, v.size()) * sizeof(T))); if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 1414;
    const int grid = argc > 2 ? atoi(argv[2]) : 2048;
    int nn = (nx + 1) * (nx + 1);
    std::vector<double> xyz;
    std::vector<int32_t> tri;
    if (argc > 3) { // a mesh file: int64 n_nodes, int64 n_tri, doubles xyz, int32 triangles (tools/lab/write_mesh.py)
        FILE *f = fopen(argv[3], "rb");
        long long hdr[2];
        if (!f || fread(hdr, 8, 2, f) != 2) { printf("cannot read %s\n", argv[3]); return 1; }
        nn = (int)hdr[0];
        xyz.resize(3 * (size_t)nn); tri.resize(3 * (size_t)hdr[1]);
        if (fread(xyz.data(), 8, xyz.size(), f) != xyz.size() || fread(tri.data(), 4, tri.size(), f) != tri.size()) { printf("short file\n"); return 1; }
        fclose(f);
    } else {
        xyz.resize(3 * (size_t)nn);
        for (int j = 0; j <= nx; j++) for (int i = 0; i <= nx; i++) { size_t a = (size_t)j * (nx + 1) + i; xyz[3*a] = 10.0 * i / nx; xyz[3*a+1] = 10.0 * j / nx; xyz[3*a+2] = 0.0; }
        tri.reserve(6 * (size_t)nx * nx);
        for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) { int n = i + j * (nx + 1), up_ = nx + 1; tri.insert(tri.end(), {n, n + 1, n + up_, n + 1, n + up_ + 1, n + up_}); }
    }
    Plan p; std::string err;
    if (!build_plan(nn, xyz.data(), (int)tri.size() / 3, tri.data(), 0, nullptr, 0, 1, &p, &err, default_symmetric_storage())) { printf("plan: %s\n", err.c_str()); return 1; }
    DeviceMatrix m;
    m.n_own = p.n_own; m.n_pad = p.n_pad; m.n_ghost = p.n_ghost; m.n_slices = p.n_slices; m.n_ltri = p.n_ltri(); m.n_lquad = 0;
    m.xyz = up(p.xyz_local); m.tri = up(p.tri_local); m.slice_width = up(p.slice_width); m.slice_base = up(p.slice_base);
    m.cols = up(p.cols); m.pair_ptr = up(p.pair_ptr); m.slice_elem_ptr = up(p.slice_elem_ptr);
    m.slice_elem_nodes = reinterpret_cast<const int4 *>(up(p.slice_elem_nodes)); m.max_slice_elems = p.max_slice_elems; m.item_ptr = up(p.item_ptr);
    m.slice_desc = reinterpret_cast<const int4 *>(up(p.slice_desc));
    m.items = reinterpret_cast<const uint4 *>(up(p.items)); m.max_stage_rows = p.max_stage_rows;
    {
        int32_t max_items = 0;
        for (int32_t s = 0; s < p.n_slices; s++) max_items = std::max(max_items, p.item_ptr[s + 1] - p.item_ptr[s]);
        (void)max_items;
        assemble_lds_layout(m, p.max_slice_elems, p.max_stage_rows, false);
    }
    std::vector<uint8_t> dm(p.n_local_nodes(), 0); m.dmask = up(dm);
    std::vector<uint32_t> fl(p.items.size(), 0u); m.item_flags = up(fl); // no Dirichlet nodes in the lab mesh
    std::vector<int32_t> st(1, 0); m.status = up(st);
    if (getenv("LAB_RHS")) { // the right-hand side beside K, as the library assembles it
        std::vector<double> ld((size_t)p.n_pad * 6, 1.0);
        m.rhs_loads = up(ld);
        double *F; CK(hipMalloc(&F, (size_t)p.n_pad * 6 * 8)); m.rhs_F = F;
    }
    double *vals; CK(hipMalloc(&vals, (size_t)p.total_slots() * 36 * 8)); m.vals = vals;
    MatConst mc; const double nu = 0.3, E = 1e7, t = 0.5;
    mc.cm = E / (1 - nu * nu); mc.cp = E * t * t * t / (12 * (1 - nu * nu)); mc.nu = nu; mc.g = (1 - nu) / 2; mc.t = t; mc.flags = 3; mc.pad = 0;
    printf("nx=%d slices=%d max_elems=%d max_stage=%d items=%zu lds=%d grid=%d\n", nx, p.n_slices, p.max_slice_elems, p.max_stage_rows, p.items.size(), m.lds_bytes, grid);
    const int R = 5;
    if (p.pipe) { // the plan's items are laid out for the pipelined kernel: nothing else can run on them
        m.pipe = 1;
        m.slice_elem_ptr_last = (int32_t)(p.slice_elem_nodes.size() / 4);
        assemble_lds_layout(m, p.max_slice_elems, 0, false);
        auto runp = [&](auto kernel, int g, int reps) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, m.lds_bytes));
            for (int i = 0; i < 25; i++) hipLaunchKernelGGL(kernel, dim3(g), dim3(256), m.lds_bytes, 0, m, mc);
            CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kernel, dim3(g), dim3(256), m.lds_bytes, 0, m, mc);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            return ms / reps;
        };
        printf("pipe lds=%d\n", m.lds_bytes);
        for (int g : {512, 256, 1024, 2048}) printf("pipe grid %4d          : %.3f ms\n", g, runp(k_assemble_pipe<0>, g, R));
        unsigned long long *st; const int g = 512;
        CK(hipMalloc(&st, (size_t)g * 4 * 8 * 8 + 16)); CK(hipMemset(st, 0, (size_t)g * 4 * 8 * 8 + 16));
        m.stamps = st;
        printf("pipe, no record math   : %.3f ms\n", runp(k_assemble_pipe<8>, g, R));
        printf("pipe, roles by wave id : %.3f ms\n", runp(k_assemble_pipe<16>, g, R));
        printf("pipe, no priority      : %.3f ms\n", runp(k_assemble_pipe<64>, g, R));
        printf("pipe, neighbours swapped: %.3f ms\n", runp(k_assemble_pipe<128>, g, R));
        printf("pipe again             : %.3f ms\n", runp(k_assemble_pipe<0>, g, R));
        const bool fake = getenv("LAB_FAKE_RECORDS") != nullptr;
        const float ms = fake ? runp(k_assemble_pipe<40>, g, 1) : runp(k_assemble_pipe<32>, g, 1);
        std::vector<unsigned long long> h((size_t)g * 4 * 8);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        printf("stamped pipe: %.3f ms; cycles per slice by role (0: diagonal items, 1-2: off-diagonal items, 3: records):\n", ms);
        for (int wv = 0; wv < 4; wv++) {
            double tw[8] = {0}, allw = 0;
            for (size_t b_ = 0; b_ < (size_t)g; b_++) for (int q = 0; q < 8; q++) { tw[q] += (double)h[(b_ * 4 + wv) * 8 + q]; allw += (double)h[(b_ * 4 + wv) * 8 + q]; }
            printf("  role %d:", wv);
            for (int q = 0; q < 7; q++) printf(" %6.0f", tw[q] / (double)p.n_slices);
            printf("  (records, barrier, block math, lane sums, stores, top, prefetch)  total %.0f\n", allw / (double)p.n_slices);
        }
        return 0;
    }
    printf("W2 full               : %.3f ms\n", run<2, 0>(m, mc, grid, R));
    printf("W3 full               : %.3f ms\n", run<3, 0>(m, mc, grid, R));
    printf("W2 plain stores       : %.3f ms\n", run<2, 64>(m, mc, grid, R));
    printf("W2 full again         : %.3f ms\n", run<2, 0>(m, mc, grid, R));
    printf("W2 no global stores   : %.3f ms\n", run<2, 1>(m, mc, grid, R));
    printf("W2 all lanes record 0 : %.3f ms\n", run<2, 2>(m, mc, grid, R));
    printf("W2 rec0 + no stores   : %.3f ms\n", run<2, 3>(m, mc, grid, R));
    printf("W2 no block math      : %.3f ms\n", run<2, 4>(m, mc, grid, R));
    printf("W2 no record math     : %.3f ms\n", run<2, 8>(m, mc, grid, R));
    printf("W2 no stores+block    : %.3f ms\n", run<2, 5>(m, mc, grid, R));
    printf("W2 skeleton (1+4+8)   : %.3f ms\n", run<2, 13>(m, mc, grid, R));
    printf("W2 no block+record    : %.3f ms\n", run<2, 12>(m, mc, grid, R));
    auto stamps = [&](auto tag, const char *title) {
        constexpr int F = decltype(tag)::value;
        unsigned long long *st; CK(hipMalloc(&st, (size_t)grid * 4 * 8 * 8 + 16)); CK(hipMemset(st, 0, (size_t)grid * 4 * 8 * 8 + 16));
        m.stamps = st;
        const float ms = run<2, F>(m, mc, grid, 1); // every launch overwrites the stamp array: one launch's data
        std::vector<unsigned long long> h((size_t)grid * 4 * 8);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        const char *names[8] = {"phase A (record math + LDS)", "barrier after A", "block math + single-item stores", "staging + barrier",
                                "reduction + diagonal stores", "top of slice (operand wait, rhs)", "prefetch issue", "(unused)"};
        double tot[8] = {0}; double all = 0;
        for (size_t w_ = 0; w_ < (size_t)grid * 4; w_++) for (int q = 0; q < 8; q++) { tot[q] += (double)h[w_ * 8 + q]; all += (double)h[w_ * 8 + q]; }
        unsigned long long clk[2]; CK(hipMemcpy(clk, st + (size_t)grid * 32, 16, hipMemcpyDeviceToHost));
        printf("in-kernel clock: %llu shader cycles per %llu ticks of 100 MHz -> %.3f GHz\n", clk[0], clk[1], 0.1 * (double)clk[0] / (double)clk[1]);
        printf("%s: %.3f ms per launch; share of wave cycles per phase:\n", title, ms);
        for (int q = 0; q < 7; q++) printf("  %-32s %5.1f %%   (%.0f cycles per wave per slice)\n", names[q], 100.0 * tot[q] / all, tot[q] / ((double)p.n_slices * 4));
        // the waves of a workgroup do different work (first wave: diagonal items; last waves: element records)
        for (int wv = 0; wv < 4; wv++) {
            double tw[8] = {0}; double allw = 0;
            for (size_t b_ = 0; b_ < (size_t)grid; b_++) for (int q = 0; q < 8; q++) { tw[q] += (double)h[(b_ * 4 + wv) * 8 + q]; allw += (double)h[(b_ * 4 + wv) * 8 + q]; }
            printf("  wave %d: cycles per slice:", wv);
            for (int q = 0; q < 7; q++) printf(" %6.0f", tw[q] / (double)p.n_slices);
            printf("  (A, barrier, block, staging+barrier, reduce, top, prefetch)  total %.0f\n", allw / (double)p.n_slices);
        }
    };
    stamps(std::integral_constant<int, 32>(), "stamped build");
    stamps(std::integral_constant<int, 32 + 13>(), "stamped skeleton (no stores, no block math, no record math)");
    stamps(std::integral_constant<int, 32 + 1>(), "stamped, no global stores");
    return 0;
}
