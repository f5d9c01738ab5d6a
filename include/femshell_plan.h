/*
 * femshell_plan.h -- host-only inspection of libfemshell's symbolic phase (row partition,
 * block sparsity, gather lists, halo lists).  No GPU is touched: these entry points exist so
 * that the partition / halo logic of the multi-GPU path can be tested on CPU-only machines
 * (tests/test_plan_cpu.py, tests/test_partition_gloo.py).  They are not part of the drop-in
 * boundary; the reference's counterpart is libMesh's DofMap/sparsity build inside
 * EquationSystems::init (fem-shell.cpp:125) and PETSc's parallel Mat/Vec layout.
 */
#ifndef FEMSHELL_PLAN_H
#define FEMSHELL_PLAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct femshell_plan femshell_plan;

int femshell_plan_create(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                         int32_t n_quad, const int32_t *quad, int32_t rank, int32_t world_size,
                         femshell_plan **out);
void femshell_plan_destroy(femshell_plan *plan);

enum {
    FEMSHELL_PLAN_N_OWN = 0, FEMSHELL_PLAN_N_PAD, FEMSHELL_PLAN_N_GHOST, FEMSHELL_PLAN_N_SLICES,
    FEMSHELL_PLAN_N_LTRI, FEMSHELL_PLAN_N_LQUAD, FEMSHELL_PLAN_TOTAL_SLOTS, FEMSHELL_PLAN_N_PAIRS,
    FEMSHELL_PLAN_N_PEERS, FEMSHELL_PLAN_ROW_BEGIN, FEMSHELL_PLAN_ROW_END, FEMSHELL_PLAN_NNZ_BLOCKS,
    FEMSHELL_PLAN_N_INTERIOR_SLICES, /* slices that read no ghost column: multiplied while the halo is in flight */
    FEMSHELL_PLAN_N_ITEMS,           /* assembly work items (<= 3 element contributions each) */
    FEMSHELL_PLAN_N_MULTI_ROUND_SLICES, /* slices with more than 256 work items (several assembly rounds) */
    FEMSHELL_PLAN_MAX_SLICE_ELEMS,   /* most elements any slice touches (LDS records) */
    FEMSHELL_PLAN_MAX_SLICE_WIDTH,   /* widest slice (block slots per node row) */
    FEMSHELL_PLAN_SYMMETRIC,         /* 1: symmetric storage -- of an off-diagonal pair of owned nodes only the block of
                                        the lower-numbered row has a slot (FEMSHELL_SYMMETRIC=0 turns it off) */
    FEMSHELL_PLAN_STORED_BLOCKS,     /* blocks that have a slot (NNZ_BLOCKS counts the blocks of the owned rows of K) */
    FEMSHELL_PLAN_PIPE,              /* 1: the work items are laid out for the pipelined assembly kernel -- rounds of 192
                                        lanes, the chunks of a slot in neighbouring lanes of one wave (csrc/plan.cpp
                                        pack_items_pipe; FEMSHELL_ASM_PIPE=0 turns it off); N_MULTI_ROUND_SLICES then counts
                                        the slices with more than 192 items */
    FEMSHELL_PLAN_INFO_COUNT
};
/* fills info[FEMSHELL_PLAN_INFO_COUNT] */
int femshell_plan_info(const femshell_plan *plan, int64_t *info);

enum {
    FEMSHELL_PLAN_GHOST_GLOBAL = 0, /* int32 [n_ghost]       global ids of the ghost nodes        */
    FEMSHELL_PLAN_TRI_LOCAL,        /* int32 [n_ltri*3]      local node ids                        */
    FEMSHELL_PLAN_TRI_GLOBAL_ID,    /* int32 [n_ltri]                                              */
    FEMSHELL_PLAN_QUAD_LOCAL,       /* int32 [n_lquad*4]                                           */
    FEMSHELL_PLAN_QUAD_GLOBAL_ID,   /* int32 [n_lquad]                                             */
    FEMSHELL_PLAN_SLICE_WIDTH,      /* int32 [n_slices]                                            */
    FEMSHELL_PLAN_SLICE_BASE,       /* int64 [n_slices+1]                                          */
    FEMSHELL_PLAN_COLS,             /* int32 [total_slots]   local column node per slot            */
    FEMSHELL_PLAN_PAIR_PTR,         /* int32 [total_slots+1]                                       */
    FEMSHELL_PLAN_PAIRS,            /* uint32 [n_pairs]      (local element<<4)|(ia<<2)|ib         */
    FEMSHELL_PLAN_XYZ_LOCAL,        /* double [(n_pad+n_ghost)*3]                                  */
    FEMSHELL_PLAN_PEER_RANKS,       /* int32 [n_peers]                                             */
    FEMSHELL_PLAN_PEER_RECV_OFFSET, /* int32 [n_peers]       first ghost index received from peer  */
    FEMSHELL_PLAN_PEER_RECV_COUNT,  /* int32 [n_peers]                                             */
    FEMSHELL_PLAN_PEER_SEND_PTR,    /* int32 [n_peers+1]     offsets into PEER_SEND_NODES          */
    FEMSHELL_PLAN_PEER_SEND_NODES,  /* int32 [sum]           owned local nodes sent to each peer   */
    FEMSHELL_PLAN_SPMV_ORDER,       /* int32 [n_slices]      interior slices first, then boundary  */
    FEMSHELL_PLAN_IN_WIDTH,         /* int32 [n_slices]      symmetric storage: transposed blocks per row  */
    FEMSHELL_PLAN_IN_BASE,          /* int64 [n_slices+1]                                          */
    FEMSHELL_PLAN_IN_SLOTS,         /* int32 [in_base.back()] slot index of a block (a, this row) or -1 */
    /* transposed products that stay inside a slice go through LDS in the SpMV kernel (csrc/plan.hpp): */
    FEMSHELL_PLAN_GAT_SLOTS,        /* int32 [in_base.back()] IN_SLOTS without the blocks of the row's own slice (-1 there) */
    FEMSHELL_PLAN_LOC_LIST,         /* uint8 [in_base.back()] position of the block among its slice's in-slice blocks, 255 = none */
    FEMSHELL_PLAN_LOC_INDEX,        /* uint8 [total_slots]    the same position per slot, 255 = the product leaves the slice */
    /* assembly work items (csrc/plan.hpp Plan::Item): four words each */
    FEMSHELL_PLAN_ITEM_PTR,         /* int32 [n_slices+1]                                          */
    FEMSHELL_PLAN_ITEMS,            /* uint32 [n_items*4]     x = slot in slice | chunk << 16 | chunks << 24, y, z = up to three
                                                              16-bit contributions and their count << 16, w          */
    FEMSHELL_PLAN_PAIRS16,          /* uint16 [n_pairs]       PAIRS with the element as index into the slice's element list */
    FEMSHELL_PLAN_SLICE_ELEM_PTR    /* int32 [n_slices+1]     range of a slice in its element list              */
};
/* returns the element count of the array; copies it to out when out != NULL */
int64_t femshell_plan_array(const femshell_plan *plan, int which, void *out);

/* Area-weighted unit normals of the plan's owned nodes (n_own x 3; what the multigrid setup projects the rotation modes with):
 * from_gather_lists = 0 walks all elements per range of nodes (rounds 3-5), 1 reads every node's elements from the gather list of
 * its diagonal slot (round 6).  The two arrays are equal bit for bit. */
int femshell_plan_node_normals(const femshell_plan *plan, int32_t from_gather_lists, double *normals_out);

/* ---- host-only pieces of the multigrid setup (csrc/amg.hpp), for CPU tests against the numpy restatement
 * oracle/amg_oracle.py.  One coarsening step: aggregation, tentative prolongator from the near-null space B
 * (n x 6 x 6: dof x mode), prolongator smoothing with omega = 4 / (3 lambda_max), Galerkin product. ---- */
int femshell_amg_host_rbm(int32_t n_nodes, const double *xyz, const uint8_t *dmask, double *B_out);

typedef struct femshell_amg_coarsening femshell_amg_coarsening;
int femshell_amg_host_coarsen(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                              const double *B, double lambda_max, femshell_amg_coarsening **out);
void femshell_amg_coarsening_destroy(femshell_amg_coarsening *h);
enum {
    FEMSHELL_COARSEN_AGG = 0,  /* int32 [n_nodes]                */
    FEMSHELL_COARSEN_P_ROWPTR, /* int64 [n_nodes+1]              */
    FEMSHELL_COARSEN_P_COLS,   /* int32                          */
    FEMSHELL_COARSEN_P_VALS,   /* double [blocks*36]             */
    FEMSHELL_COARSEN_AC_ROWPTR,/* int64 [n_coarse+1]             */
    FEMSHELL_COARSEN_AC_COLS,
    FEMSHELL_COARSEN_AC_VALS,
    FEMSHELL_COARSEN_BC        /* double [n_coarse*36]: coarse near-null space */
};
int64_t femshell_amg_coarsening_array(const femshell_amg_coarsening *h, int which, void *out);
/* the aggregation alone (graph of the block pattern).  visit: NULL, or the order in which the greedy passes meet the nodes
 * (a permutation of 0..n-1) -- what femshell_set_mesh hands over when it renumbered the nodes itself, so that the
 * aggregates are those of the caller's numbering.  Returns the number of aggregates (< 0: invalid argument). */
int32_t femshell_amg_host_aggregate(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const int32_t *visit,
                                    int32_t *agg_out);
/* the patch smoother's clusters of rigidly coupled nodes on a host matrix (csrc/amg_patch.hpp; the device finds them with the same
 * arithmetic): labels_out[i] = cluster of node i or -1, clusters numbered by their smallest node.  Returns the number of clusters
 * (< 0: invalid argument); *edges_out (may be NULL) = rigid edges found. */
int32_t femshell_amg_host_patch_clusters(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals, double tau,
                                         int32_t max_nodes, int32_t *labels_out, int64_t *edges_out);
/* the aggregation with every cluster (labels: cluster of a node or -1) glued into one node first */
int32_t femshell_amg_host_aggregate_glued(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const int32_t *labels,
                                          const int32_t *visit, int32_t *agg_out);
/* dense inverse of a small SPD block matrix (coarsest level): inv_out (6n)^2 doubles */
int femshell_amg_host_dense_inverse(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                    double *inv_out);
/* the sliced block ELL image the device kernels read, for layout tests: returns total slots; arrays may be NULL */
int64_t femshell_amg_host_pack(int32_t n_rows, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                               int32_t diag_first, int32_t *slice_width, int64_t *slice_base, int32_t *cols, double *ell_vals);

/* the node ordering FEMSHELL_REORDER_MORTON (kind 0) / FEMSHELL_REORDER_RCM (kind 1) would use inside femshell_set_mesh:
 * perm_out[new index] = caller's node id (csrc/reorder.cpp) */
int femshell_reorder_host(int32_t kind, int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                          int32_t n_quad, const int32_t *quad, int32_t *perm_out);

/* symmetric storage of a square operator (coarse multigrid levels): diagonal + upper blocks in the sliced ELL arrays and
 * the in-lists of the transposed products; returns the total slots, *in_total receives the in-list entries; arrays may
 * be NULL for a sizing call */
int64_t femshell_amg_host_pack_sym(int32_t n_rows, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                   int32_t *slice_width, int64_t *slice_base, int32_t *cols, double *ell_vals,
                                   int32_t *in_width, int64_t *in_base, int32_t *in_slots, int32_t *in_rows, int64_t *in_total);

#ifdef __cplusplus
}
#endif
#endif
