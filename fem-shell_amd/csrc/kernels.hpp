// kernels.hpp -- host-callable launchers of the gfx950 kernels (defined in kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shell_element.hpp"

namespace femshell {

constexpr int32_t kStatusDirect = 0x40000000; // status values above this carry a local element id directly

// Device view of the mesh + matrix structure of one rank (see plan.hpp for the layout).
struct DeviceMatrix {
    int32_t n_own = 0, n_pad = 0, n_ghost = 0, n_slices = 0;
    int32_t n_ltri = 0, n_lquad = 0;
    const double *xyz = nullptr;        // (n_pad+n_ghost) x 3
    const int32_t *tri = nullptr;       // n_ltri x 3, local node ids
    const int32_t *quad = nullptr;      // n_lquad x 4
    const int32_t *slice_width = nullptr;
    const int64_t *slice_base = nullptr;
    const int32_t *cols = nullptr;      // per slot
    const int32_t *pair_ptr = nullptr;  // per slot + 1
    const int32_t *slice_elem_ptr = nullptr; // n_slices+1
    const int4 *slice_desc = nullptr;        // 2 per slice: {elem begin, elem count, item begin, item count},
                                             // {slot base lo, hi, width, 0} -- what k_assemble needs of a slice
    const int4 *slice_elem_nodes = nullptr;  // per slice element: its local node ids (w = -1 for TRI3)
    int32_t max_slice_elems = 0;
    int32_t max_slice_width = 0;             // widest slice (block slots per node row)
    const int32_t *item_ptr = nullptr;       // n_slices+1
    const uint4 *items = nullptr;            // assembly work items (plan.hpp)
    uint32_t *item_flags = nullptr;          // per item: what its owner lane needs to know about the slot besides the
                                             // contributions (k_item_flags): Dirichlet mask of the row node (bits 0-5)
                                             // and of the column node (6-11), contributions in the slot (12-19), bit 20
                                             // = diagonal slot; 0 for items that do not own their slot
    int32_t max_stage_rows = 0;
    int32_t lds_bytes = 0;                   // dynamic LDS of k_assemble (assemble_lds_layout)
    int32_t lds_rec_off = 0, lds_stage_off = 0; // offsets in doubles (pipe: lds_rec_off = doubles of one record buffer)
    int32_t pipe = 0;                        // items laid out for k_assemble_pipe (Plan::pipe)
    int32_t slice_elem_ptr_last = 0;         // entries of slice_elem_nodes
    const uint8_t *dmask = nullptr;     // per local node (owned, padding, ghosts)
    double *vals = nullptr;             // total_slots x 36, sliced layout
    const float *vals32 = nullptr;      // the same in single precision (smoothing products of the multigrid cycle only)
    // ... and what such a product (symmetric storage) keeps in single precision besides: 1 = its results -- the direct part
    // y and the transposed products in tbuf, six floats where the FP64 product writes six doubles, in the same buffers --,
    // 2 = its input vector as well.  The arithmetic stays FP64.  k_cheb_start / k_cheb_step / k_sym_gather are told the
    // same number (amg_solve.cpp Cycle::smooth).
    int32_t vec32 = 0;
    const double *rhs_loads = nullptr;  // n_pad x 6 nodal loads and
    double *rhs_F = nullptr;            // the right-hand side k_assemble fills beside K (nullptr: K only)
    // symmetric storage (plan.hpp): transposed products K_ac^T x_a next to every slot, collected per row
    int32_t symmetric = 0;
    // ... and of a diagonal block (slot 0 of a row, symmetric itself) only the 12 of its 18 words that hold the upper
    // triangle are ever WRITTEN: word (jp, i) carries the columns 2jp, 2jp+1 of row i and lies below the diagonal when
    // 2jp+1 < i.  The assembly kernels skip those six stores (192 of 288 bytes per node reach HBM), the other six words of
    // the slot are never-written memory, and every reader of a diagonal slot takes (i, j), j < i, from (j, i): k_spmv_sym,
    // k_residual_dd_sym, k_block_jacobi, the multigrid setup's block loads, the exports.  K only (1 with symmetric storage
    // unless FEMSHELL_DIAG_UPPER=0); the level operators of the multigrid write whole blocks.
    int32_t diag_upper = 0;
    int32_t max_in_width = 0;
    const int32_t *in_width = nullptr;  // n_slices
    const int64_t *in_base = nullptr;   // n_slices+1
    const int32_t *in_slots = nullptr;  // per in-entry: slot index or -1
    const int32_t *in_rows = nullptr;   // per in-entry: local row of that block
    // products that stay inside a slice go through LDS in k_spmv_sym (plan.hpp); loc_index == nullptr: all through tbuf
    const int32_t *gat_slots = nullptr; // in_slots without the in-slice entries: what the collecting kernels walk
    const uint8_t *loc_index = nullptr; // per slot
    const uint8_t *loc_list = nullptr;  // per in-entry
    int32_t max_loc = 0;
    double *tbuf = nullptr;             // total_slots x 6: per (slice, slot k) [component j][node n]
    double *minv = nullptr;             // n_slices x 21 x 32: upper triangles of the inverse diagonal blocks
    const float *minv32 = nullptr;      // the same in single precision: the Chebyshev smoothers of the multigrid cycle read it
                                        // when set (amg_solve.cpp); the CG's own block-Jacobi step never does
    unsigned long long *stamps = nullptr; // profiling builds of k_assemble only (tools/lab)
    int32_t *status = nullptr;          // device int: 0 ok, e+1 = first degenerate local element,
                                        // -(node+1) = singular diagonal block
};

// Record i of a slice in a record buffer of k_assemble_pipe.  34 (66) doubles per record are 68 (132) dwords = 4 mod 64
// banks: records whose indices differ by 16 start in the same bank, and on a structured slice the lanes n and n + 8 of a wave
// read exactly such records (node n's elements sit at 2n + const in the slice's list) -- a four-way conflict per 32-lane
// group.  Two doubles of padding behind every 16 records move them four banks apart; the records stay 16-byte aligned.
// MEASURED (round 4, 4M-triangle panel, alternating libraries on one box): 0.550-0.552 ms with the padding against
// 0.543-0.552 ms without -- the conflicts the counters show (33 % of the LDS-active cycles) are not on the critical path.
// Off by default; -DFEMSHELL_PIPE_REC_PAD=2 builds the padded layout.
#ifndef FEMSHELL_PIPE_REC_PAD
#define FEMSHELL_PIPE_REC_PAD 0
#endif
constexpr int kPipeRecPad = FEMSHELL_PIPE_REC_PAD;
__host__ __device__ constexpr int pipe_rec_offset(int i, int rec_doubles) { return i * rec_doubles + kPipeRecPad * (i >> 4); }

// LDS layout of k_assemble for a plan with at most max_slice_elems element records and max_stage_rows partial-sum
// rows per slice: [records | partial sums].  Returns the dynamic LDS size in bytes.
inline size_t assemble_lds_layout(DeviceMatrix &m, int32_t max_slice_elems, int32_t max_stage_rows, bool has_quads)
{
    if (m.pipe) { // k_assemble_pipe: two buffers of lean records, no staging
        // (+ two doubles of padding behind every 16 records: pipe_rec_offset)
        m.lds_rec_off = pipe_rec_offset(max_slice_elems, has_quads ? kRecDoublesQuad : RecLean::doubles);
        m.lds_stage_off = 0;
        m.lds_bytes = (int32_t)(2 * (size_t)m.lds_rec_off * sizeof(double));
        return (size_t)m.lds_bytes;
    }
    const int rec = has_quads ? kRecDoublesQuad : kRecDoubles;
    m.lds_rec_off = 0;
    m.lds_stage_off = max_slice_elems * rec;
    m.lds_bytes = (int32_t)(((size_t)max_slice_elems * rec + (size_t)max_stage_rows * 36) * sizeof(double));
    return (size_t)m.lds_bytes;
}
// Scalars of the CG recurrence, resident in HBM (no host round trip per iteration).
struct CgScalars {
    double rz;     // r.z of the current iterate
    double alpha;
    double beta;
    double rr;     // r.r
    double bb;     // b.b
    double tol2;   // rtol^2 * b.b
    double red[3]; // local sums before / global sums after the all-reduce
    int32_t done;  // 0 running, 1 converged, -1 breakdown
    int32_t iters; // iterations performed
    uint32_t ticket;     // arrival counter of the two-stage reduction (0 between launches)
    uint32_t pad;
    double stage[3][64]; // stage-1 sums of the reduction workgroups
    // single-reduction recurrence with the scalar step folded into the update kernel (k_cgcg_update): every workgroup
    // derives alpha / beta from red[] and the previous (r.z, alpha); those are read from ring[parity] while workgroup 0
    // writes ring[parity ^ 1] and the fields above
    double ring_rz[2], ring_alpha[2];
    // a refinement pass of the multigrid-preconditioned solve (cg_amg): ||x||^2 of the iterate the pass corrects (uploaded by the
    // host in front of CG_PHASE_FLEX_RESTART; 0: no adaptive stopping), ||rhs||^2 of the pass, the tolerance of the solve
    double pass_xx, pass_rhs_rr, pass_rtol;
};

struct CgVectors {
    double *x = nullptr;  // solution, n_pad*6
    double *r = nullptr;  // residual
    double *z = nullptr;  // preconditioned residual
    double *p = nullptr;  // search direction, (n_pad+n_ghost)*6 (ghost part filled by the halo exchange)
    double *q = nullptr;  // A p (single-reduction recurrence: w = A z)
    double *sv = nullptr; // single-reduction recurrence only: s = A p by recurrence
    const double *b = nullptr; // right-hand side
    double *partials = nullptr; // 2 x grid doubles
    CgScalars *s = nullptr;
    double *hist = nullptr; // r.r / b.b per iteration
    int32_t hist_cap = 0;
};

enum CgPhase : int {
    CG_PHASE_NONE = 0, CG_PHASE_INIT = 1, CG_PHASE_ALPHA = 2, CG_PHASE_BETA = 3, CG_PHASE_RESTART = 4,
    CG_PHASE_FUSED_INIT = 5, CG_PHASE_FUSED_STEP = 6, // single-reduction (Chronopoulos-Gear) recurrence
    // flexible PCG around a general preconditioner (multigrid cycle, amg_solve.cpp):
    CG_PHASE_FLEX_INIT = 7, // red[0] = b.b
    CG_PHASE_FLEX_RZ0 = 8,  // red[0] = r.z of the initial residual
    CG_PHASE_FLEX_CONV = 9, // red[0] = r.r after the update: iteration count, history, stopping test
    CG_PHASE_FLEX_BETA = 10, // red[0] = r.z, red[1] = z.q: beta = z.(r - r_old) / rz_old = -alpha z.q / rz_old
    CG_PHASE_FLEX_RESTART = 11, // red[0] = r.r of the right-hand side of a refinement pass (the double-double residual
                                // of the accumulated solution); tolerance, iteration count and history carry on
    CG_PHASE_FLEX_WARM = 12     // a solve from an initial guess: red[0] = r.r of b - K x0 (double-double), the first phase solves the
                                // correction equation down to the threshold CG_PHASE_FLEX_INIT derived from b
};

int slice_grid(const DeviceMatrix &m); // workgroups of the per-slice kernels (multiple of 8, at most 2560)

void launch_assemble(const DeviceMatrix &m, const MatConst &mc, hipStream_t st);
void launch_rhs(const DeviceMatrix &m, const double *loads /* n_pad x 6 */, double *F, hipStream_t st);
// fills m.item_flags from cols / dmask / pair_ptr (after every change of the Dirichlet set)
void launch_item_flags(const DeviceMatrix &m, int64_t n_items, hipStream_t st);
void launch_block_jacobi(const DeviceMatrix &m, hipStream_t st);
// agree[0] = 1 if *status > 0 (degenerate element), agree[1] = 1 if *status < 0 (singular diagonal block): the counters the
// ranks of a row partition sum up after an assembly / block-Jacobi setup
void launch_status_flags(const int32_t *status, double *agree, hipStream_t st);
void launch_element_matrices(const DeviceMatrix &m, const MatConst &mc, int32_t first, int32_t count,
                             double *Ke_out, hipStream_t st);

// y = K x; when partials != nullptr also partials[wg] = sum over the workgroup's rows of x*y
// (s != nullptr: no-op once s->done != 0)
void launch_spmv(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                 hipStream_t st);
// y = base_vec + sign * K x (base_vec may be y itself): residuals b - K x and prolongations x + P x_c of the
// multigrid cycle; K may be rectangular (x indexed by the block columns, y and base_vec by the block rows)
// One Chebyshev step of the smoother in D^-1 A on a full-storage operator, fused with its product (multigrid levels that
// are bound by the latency between launches): r_out = r_in - A d_in; d_out = a d_in + c D^-1 r_out; x += d_out.
// d_out must not be d_in (other workgroups still read d_in); r_out may be r_in.
void launch_spmv_cheb(const DeviceMatrix &m, const double *d_in, const double *r_in, double *r_out, double *d_out, double *x,
                      double a, double c, const CgScalars *s, hipStream_t st);
// The residual in front of a post-smoothing and the smoothing's FIRST step in one launch (full storage): r_out = r_in - A v_in;
// d_out = inv_theta D^-1 r_out; x += d_out  (d_out must not be v_in; r_out may be r_in)
void launch_spmv_start(const DeviceMatrix &m, const double *v_in, const double *r_in, double *r_out, double *d_out, double *x,
                       double inv_theta, const CgScalars *s, hipStream_t st);
void launch_spmv_axpy(const DeviceMatrix &m, const double *x, double *y, const double *base_vec, double sign,
                      const CgScalars *s, hipStream_t st);
// full storage only: the same, and the product itself (K x, without the base vector) into prod_out -- as floats in that buffer
// when prod_float (the coarse correction P x_c, which the cycle adds to its iterate and then multiplies with the level operator)
void launch_spmv_axpy_keep(const DeviceMatrix &m, const double *x, double *y, const double *base_vec, double sign, double *prod_out,
                           bool prod_float, const CgScalars *s, hipStream_t st);
// symmetric storage only: second phase of a product whose first phase ran through launch_spmv_span (the transposed
// products of all slices must be in place): y = base_vec + sign * (y + sum of the row's transposed products)
// (q32: y and the transposed products come from a smoothing product that stored them in single precision -- DeviceMatrix::vec32 --
//  and the result goes to `out`, FP64, which may be base_vec)
void launch_sym_gather(const DeviceMatrix &m, double *y, const double *base_vec, double sign, const CgScalars *s, hipStream_t st,
                       bool q32 = false, double *out = nullptr);
// r = b - K x with double-double products and row sums (accurate residual for the residual replacement; r != x)
void launch_residual_dd(const DeviceMatrix &m, const double *x, const double *b, double *r, hipStream_t st);
// what a full-storage product does with its result besides storing it (the epilogues of launch_spmv_axpy / _cheb / _start /
// _axpy_keep as one description), for launch_spmv_epilogue_span
struct SpmvEpilogue {
    const double *base_vec = nullptr; // y = base_vec + sign * K x (nullptr: y = K x)
    double sign = 1.0;
    double *d_out = nullptr, *xsol = nullptr; // Chebyshev step: d_out = a x + c D^-1 y, xsol += d_out (start != 0: the first step)
    double a = 0.0, c = 0.0;
    int start = 0;
    double *prod_out = nullptr;       // K x itself as well (as floats: prod_float)
    bool prod_float = false;
};
// the product with such an epilogue over the slices order[begin, begin + count) only
void launch_spmv_epilogue_span(const DeviceMatrix &m, const double *x, double *y, const SpmvEpilogue &e, const int32_t *order, int begin,
                               int count, const CgScalars *s, hipStream_t st);
// the same over the slices order[begin, begin+count) only (interior / boundary halves of an overlapped
// halo exchange); the partial sums go to partials[partial_offset ...]; returns the number written
int launch_spmv_span(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                     const int32_t *order, int begin, int count, int partial_offset, hipStream_t st);

// CG steps; every kernel is a no-op once s->done != 0
// restart = false: x=0, r=b;  restart = true: x kept, r = b - q (q = K x computed by the caller);
// then z = M^-1 r, p = z, partial sums of r.z and (b.b | r.r)
void launch_cg_init(const DeviceMatrix &m, const CgVectors &v, bool restart, hipStream_t st);
// p[owned rows] = x (to run q = K x through the SpMV kernel, whose input carries the ghost entries)
void launch_copy_x_to_p(const DeviceMatrix &m, const CgVectors &v, hipStream_t st);
// x,r,z + partial r.z, r.r.  gather (symmetric storage): v.q holds the direct part of K p only (launch_spmv_direct or
// spans without launch_sym_gather); the kernel adds the transposed products of each row's in-list itself
void launch_cg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st, bool gather = false);
// symmetric storage: first phase of y = K x only (direct part of y, transposed products into m.tbuf, fused x.Kx sums)
void launch_spmv_direct(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s, hipStream_t st,
                        bool single_precision_values = false);
// dst = (float)src
void launch_to_f32(const double *src, float *dst, int64_t n, hipStream_t st);
// v rounded to `sig` (< 24) significant bits in place: FEMSHELL_AMG_SMOOTH_SIGBITS, an experiment knob
void launch_round_sig(float *v, int64_t n, int sig, hipStream_t st);

void launch_cg_direction(const DeviceMatrix &m, const CgVectors &v, hipStream_t st); // p = z + beta p
// single-workgroup scalar step: optional reduction of `nsums` partial arrays into s->red, then the
// scalar update of `phase` (rtol only used by CG_PHASE_INIT)
// (n_partials > 0: length of each partial array, default slice_grid(m); nsums == 3: the third array, the SpMV's,
// starts at 2 * n_partials and holds len3 entries)
// gate_phase >= 0: the phase whose skip rule applies (the reduce-only launch in front of an all-reduce runs phase
// NONE on behalf of another phase; INIT / RESTART steps must also run on a finished solve)
void launch_cg_scalar(const DeviceMatrix &m, const CgVectors &v, bool reduce, int nsums, CgPhase phase,
                      double rtol, hipStream_t st, int n_partials = 0, int len3 = 0, int gate_phase = -1);

// Single-reduction preconditioned CG (Chronopoulos & Gear): one all-reduce of (r.z, r.r, z.Az) per iteration.
//   init:   x = 0, r = b, z = M^-1 r, p = s = 0, partial sums of r.z and r.r
//   update: p = z + beta p, s = w + beta s, x += alpha p, r -= alpha s, z = M^-1 r, partial sums of r.z and r.r
// with w = A z (v.q) from the SpMV kernel, whose fused dot is z.w
void launch_cgcg_init(const DeviceMatrix &m, const CgVectors &v, hipStream_t st);
// step >= 0: the kernel first performs the scalar step that closes the previous iteration (red[] holds its global sums;
// reads ring[step & 1], writes ring[(step & 1) ^ 1]); gather: q holds the direct part of a symmetric product only, the
// rows add the transposed products of their in-lists
void launch_cgcg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st, int step = -1, bool gather = false);

// halo: gather owned entries of p into a contiguous send buffer (width doubles per node: 6 for the vectors of the solve;
// the multigrid setup of row-partitioned contexts sends rows of 1, 36 and more doubles the same way)
void launch_pack(const double *p, const int32_t *send_nodes, int32_t count, double *sendbuf, hipStream_t st, int width = 6);

} // namespace femshell
