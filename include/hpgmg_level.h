/*
 * hpgmg_level.h -- storage contract of the HPGMG-FV operator layer.
 *
 * The struct family below (blockCopy_type, communicator_type, box_type,
 * level_type) is the data contract every operators.h function works on.  It is
 * field-for-field layout compatible with the non-MPI build of the reference
 * (reference finite-volume/source/level.h:65-200) so that an operator plugin
 * written against this header can be linked under the reference's own
 * level.c / mg.c / solvers.c driver, and vice versa.  Anything this
 * implementation needs in addition (device mirrors of the block lists, the
 * transport used instead of MPI, the slab allocation) lives in a side record
 * reached through hpgmg_level_ext() -- never inside level_type.
 *
 * Memory model: for one box, vector v, cell (i,j,k) with i,j,k in [-g, dim+g):
 *     vectors[v][ (i+g) + (j+g)*jStride + (k+g)*kStride ]
 * vectors[v] = vectors[0] + v*volume.  Face-centred coefficients beta_{i,j,k}
 * store the LOW face of a cell at the cell's index (reference defines.h:35-37).
 * In the HIP build vectors[] are DEVICE pointers; the host never dereferences
 * them.
 */
#ifndef HPGMG_LEVEL_H
#define HPGMG_LEVEL_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* boundary condition kinds (reference level.h:25-26) */
#define BC_PERIODIC  0
#define BC_DIRICHLET 1

/* which parts of the 26-neighbourhood a stencil touches (reference level.h:28-31) */
#define STENCIL_SHAPE_BOX        0 /* faces + edges + corners */
#define STENCIL_SHAPE_STAR       1 /* faces only              */
#define STENCIL_SHAPE_NO_CORNERS 2 /* faces + edges           */
#define STENCIL_MAX_SHAPES       3

/* default tile of the flattened box->block list (reference level.h:34-44) */
#define BLOCKCOPY_TILE_I 10000
#define BLOCKCOPY_TILE_J 8
#define BLOCKCOPY_TILE_K 8

/* restriction flavours (reference operators.h:9-12) */
#define RESTRICT_CELL   0
#define RESTRICT_FACE_I 1
#define RESTRICT_FACE_J 2
#define RESTRICT_FACE_K 3

/* One strided 3-D region copy / stencil tile.  box>=0 addresses
 * my_boxes[box].vectors[id] (coordinates relative to the first interior cell),
 * box<0 addresses the raw buffer ptr (i is then a linear offset). */
typedef struct {
  int subtype;                 /* BC lists: 13+di+3dj+9dk of the DOMAIN normal */
  struct { int i, j, k; } dim; /* extent of the region (in write-space units for copies) */
  struct { int box, i, j, k, jStride, kStride; double *ptr; } read, write;
} __attribute__((aligned(64))) blockCopy_type;

/* A three-phase "mini program": [0] pack into send buffers, [1] local box->box,
 * [2] unpack from receive buffers (reference level.h:77-92). */
typedef struct {
  int      num_recvs;
  int      num_sends;
  int     *recv_ranks;
  int     *send_ranks;
  int     *recv_sizes;   /* in doubles */
  int     *send_sizes;
  double **recv_buffers; /* recv_buffers[0] is the bulk allocation */
  double **send_buffers;
  int      allocated_blocks[3];
  int      num_blocks[3];
  blockCopy_type *blocks[3];
} communicator_type;

typedef struct {
  int global_box_id;               /* index into level->rank_of_box */
  struct { int i, j, k; } low;     /* global coordinate of the first interior cell */
  int dim;                         /* interior cells per side */
  int ghosts;                      /* ghost depth */
  int jStride, kStride, volume;    /* in doubles */
  int numVectors;
  double **vectors;                /* vectors[v] -> start of the padded 3-D array of vector v */
  double  *fp_base;                /* allocation the vectors were carved from (may be shared) */
} box_type;

typedef struct {
  double h;                        /* grid spacing */
  int active;                      /* this rank has work on this or a coarser level */
  int num_ranks;
  int my_rank;
  int box_dim;
  int box_ghosts;
  int box_jStride, box_kStride, box_volume;
  int numVectors;
  int tag;                         /* log2(dim.i): disambiguates messages of different levels */
  struct { int i, j, k; } boxes_in;
  struct { int i, j, k; } dim;     /* global cells per side */

  int      *rank_of_box;           /* [boxes_in.k][boxes_in.j][boxes_in.i], -1 = hole */
  int       num_my_boxes;
  box_type *my_boxes;

  int allocated_blocks;
  int num_my_blocks;
  blockCopy_type *my_blocks;       /* every owned box cut into dim x 8 x 8 tiles */

  struct {
    int type;
    int allocated_blocks[STENCIL_MAX_SHAPES];
    int num_blocks[STENCIL_MAX_SHAPES];
    blockCopy_type *blocks[STENCIL_MAX_SHAPES];
  } boundary_condition;

  communicator_type exchange_ghosts[STENCIL_MAX_SHAPES];
  communicator_type restriction[4];
  communicator_type interpolation;

  double dominant_eigenvalue_of_DinvA;
  int    must_subtract_mean;
  double *RedBlack_base;           /* kept for layout compatibility; unused (colour is computed in-kernel) */
  double *RedBlack_FP;
  double *fluxes;

  int num_threads;

  struct {
    double smooth, apply_op, residual, blas1, blas3, boundary_conditions;
    double restriction_total, restriction_pack, restriction_local, restriction_unpack,
           restriction_recv, restriction_send, restriction_wait;
    double interpolation_total, interpolation_pack, interpolation_local, interpolation_unpack,
           interpolation_recv, interpolation_send, interpolation_wait;
    double ghostZone_total, ghostZone_pack, ghostZone_local, ghostZone_unpack,
           ghostZone_recv, ghostZone_send, ghostZone_wait;
    double collectives;
    double Total;
  } timers;
  int Krylov_iterations;
  int CAKrylov_formations_of_G;
  int vcycles_from_this_level;
} level_type;

/* ---- level construction (reference level.c:1075, :929, :1265, :1305, :313) ---- */
void create_level(level_type *level, int boxes_in_i, int box_dim, int box_ghosts, int numVectors,
                  int domain_boundary_condition, int my_rank, int num_ranks);
void destroy_level(level_type *level);
void create_vectors(level_type *level, int numVectors);
void reset_level_timers(level_type *level);
void append_block_to_list(blockCopy_type **blocks, int *allocated_blocks, int *num_blocks,
                          int dim_i, int dim_j, int dim_k,
                          int read_box,  double *read_ptr,  int read_i,  int read_j,  int read_k,
                          int read_jStride,  int read_kStride,  int read_scale,
                          int write_box, double *write_ptr, int write_i, int write_j, int write_k,
                          int write_jStride, int write_kStride, int write_scale,
                          int tile_i, int tile_j, int tile_k, int subtype);

#ifdef __cplusplus
}
#endif
#endif
