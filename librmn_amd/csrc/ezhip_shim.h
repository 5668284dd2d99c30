/*
 * ezhip_shim.h -- internal boundary between the C host front-end (ez_host.c, pack_host.c) and the
 * HIP translation units (ez_kernels.hip, pack_kernels.hip).  Plain C types only.
 *
 * The host front-end never includes a HIP header: device memory, streams and kernel launches are
 * reached exclusively through these functions ("thin HIP C-ABI shim", BASELINE.json north_star).
 */
#ifndef EZHIP_SHIM_H
#define EZHIP_SHIM_H
#include <stddef.h>
#include "ezhip_develop.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime plumbing ---------------------------------------------------------------- */
int    ezhip_runtime_ok(void);                       /* 1 if a HIP device is usable, else 0 */
int    ezhip_bound_device_ok(const char *who);       /* 0 when the calling thread's current device is the one the library is bound to (binds on first use); -1 + message otherwise */
int    ezhip_bound_device(void);                     /* the bound device, -1 before the first use */
const char *ezhip_last_error(void);
void ezhip_note_error(void);
unsigned ezhip_error_count(void);      /* failed runtime calls / launches so far (process-wide) */
void  *ezhip_malloc(size_t nbytes);                  /* NULL on failure */
void   ezhip_free(void *d);
int    ezhip_h2d(void *d, const void *h, size_t nbytes);     /* async on the current stream */
int    ezhip_d2h(void *h, const void *d, size_t nbytes);     /* async on the current stream */
int    ezhip_d2d(void *dst, const void *src, size_t nbytes);
int    ezhip_h2d_blocking_own_stream(void *d, const void *h, size_t nbytes);   /* blocking, on a stream of the calling thread's own (uploader thread of the host-pointer ABI) */
void   ezhip_own_stream_release(void);
int    ezhip_d2h_pinned(void *h, const void *d, size_t nbytes);   /* into page-locked memory of the library: plain asynchronous copy */
void   ezhip_touch_writable(void *h, size_t nbytes);     /* the CPU writes every page of a host range onto itself before the device writes it */
int    ezhip_memset(void *d, int v, size_t nbytes);
int    ezhip_sync(void);                             /* synchronise the current stream */
void   ezhip_set_stream(void *hip_stream);           /* thread-local; NULL = default stream */
void  *ezhip_get_stream(void);
void  *ezhip_host_alloc(size_t nbytes);              /* pinned host staging */
void   ezhip_host_free(void *p);
int    ezhip_host_pin(void *p, size_t nbytes);       /* hipHostRegister / hipHostUnregister of a caller's array */
int    ezhip_host_unpin(void *p);
int    ezhip_device_error(void);                     /* sticky device-side error word (pinned host memory): returns and clears it */

/* ---- interpolation plans --------------------------------------------------------------- */

/* row/column flags of the separable plan */
enum {
    EZF_MAIN   = 0,
    EZF_DEHORS = 1          /* column or row lies outside the source grid: write the fill value */
};

/* A "special" target row: every tap names a source row (>= 0) or a synthetic pole row. */
enum { EZ_ROW_POLE_N = -1, EZ_ROW_POLE_S = -2 };

typedef struct {
    int    row;             /* target row (0-based) */
    int    kind;            /* 0 = strip interpolation, 1 = whole row := north pole value,
                               2 = whole row := south pole value, 3 = whole row := fill value */
    int    tap[4];          /* source rows / EZ_ROW_POLE_* */
    double w[4];            /* cubic: 4 weights; linear: w[0] = dy; nearest: unused */
} ezhip_special_row;

/*
 * Separable plan ("mode A"): source and target are both rectilinear in lat/lon, so the located
 * x depends only on the target column and y only on the target row.  All arrays are DEVICE
 * pointers, SoA with leading dimension ni_dst (columns) or nj_dst (rows).
 */
/* k_sep geometry shared by the plan builder (host) and the kernel */
#define EZHIP_SEP_COLS 256            /* target columns per block = threads per block */
#define EZHIP_SEP_ROWS 16             /* target rows per block */
#define EZHIP_SEP_WMAX 272            /* max staged source columns per block (>= 256 + stencil) */
#define EZHIP_SEP_RMAX 20             /* max staged source rows per block (>= 16 + stencil) */

/* per target row, AoS so that one row is two scalar loads; padded to a multiple of EZHIP_SEP_ROWS rows */
typedef struct {
    int    jb;              /* first source row of the taps, RELATIVE to the row-block's staged patch */
    int    flag;            /* != 0: not a main row (special row, or padding past the last row) */
    int    pad0, pad1;
    double w[4];            /* cubic: y weights; linear: w[0] = dy */
} ezhip_rowinfo;

/* one staging step of k_sepx: source rows [s0, s0 + n) are new to the ring, the first lands in slot0 */
typedef struct { int s0, n, slot0, by; } ezhip_xstep;
/* the 16 target rows of a row-block as 64-byte records (they ride the LDS-DMA; one lane of the y-pass reads the
 * record of ITS row): y weights, byte offsets of the 4 taps in the ring (wrap resolved on the host), element offset
 * of the target row to store (a row that is not a main row repeats a neighbouring main row) */
typedef struct { double w[4]; int t_off[4]; unsigned o_off; int pad[3]; } ezhip_xrow;

typedef struct {
    int degree;                       /* 0 nearest, 1 linear, 3 cubic */
    int ni_src, nj_src, ni_dst, nj_dst;
    const ezhip_rowinfo *rowinfo;     /* [nblk_y * EZHIP_SEP_ROWS] */
    int nblk_y;                       /* row-blocks of EZHIP_SEP_ROWS target rows */
    int wstride, patch_elems;         /* LDS patch row stride and size (floats) of ONE of the two buffers */
    const int    *cidx;               /* [4][ni_dst] 0-based source column of each tap */
    const int    *coff;               /* [4][ni_dst] the same taps as offsets into the block's staged patch */
    const int    *blk_base;           /* [nblk_x] first staged source column of the block, -1: patch not usable */
    const int    *blk_w;              /* [nblk_x] staged width (<= EZHIP_SEP_WMAX) */
    const int    *brow_s0;            /* [nblk_y] first staged source row of the row-block */
    const int    *brow_n;             /* [nblk_y] staged rows (<= EZHIP_SEP_RMAX), 0: not usable */
    const double *cw;                 /* [4][ni_dst] cubic weights | linear: cw[0][] = dx */
    const int    *cidx_s;             /* same, for the polar-strip kernels (wnnc / regular forms) */
    const double *cw_s;
    const unsigned char *cflag;       /* [ni_dst] EZF_* */
    const int    *rbase;              /* [nj_dst] 0-based first source row of the taps */
    const double *rw;                 /* [4][nj_dst] cubic weights | linear: rw[0][] = dy */
    const unsigned char *rflag;       /* [nj_dst] 0 = main row, 1 = handled as a special row */
    const ezhip_special_row *special; /* [n_special] */
    int n_special;
    int pole_weighted;                /* 1: Z-on-E trapezoid pole value (needs ax) */
    const float *ax;                  /* device source x axis (pole_weighted only) */
    int vector_mode;                  /* 1: strip pole rows come from pole_rows_n/s instead of a scalar */
    const float *pole_row_n, *pole_row_s;   /* [ni_src] synthetic polar wind rows (vector mode) */
    const float *fill;                /* device scalar written to DEHORS points (may be NULL) */
    const float *polevals;            /* device float[2] = {north, south} pole values of THIS field (ezhip_polevals); scalar mode */
    int need_poles;                   /* some special row uses a scalar pole value */
    int debug_flags;                  /* development only (EZHIP_DEBUG knock-outs of k_sepx): 1 no stores, 4 no DMA, 8 no x-pass, 16 no y-pass */
    /* ---- k_sepx (linear / cubic default when x_nseg > 0): one thread block = one 256-column strip x x_rb
     * consecutive row-blocks.  Every source row of the strip is staged (LDS-DMA) and x-interpolated ONCE per
     * thread block; the x-pass results live in a per-thread LDS ring of x_tr source rows (fp64), the y-pass
     * reads its four taps from the ring at a row-uniform slot. */
    int x_nseg, x_rb, x_nvb;          /* segments per strip, row-blocks per segment, valid row-blocks */
    int x_tr, x_prows, x_rows_per_step;   /* ring rows, patch rows, target rows per step (8 or 16) */
    int batch_fields;                 /* > 1: one launch covers this many fields, strides below, polevals[2 f] */
    /* in-kernel pole values (k_sepx): pole_blocks = 2 * fields rounded up to a multiple of 8 producer blocks (0: pole
     * values come precomputed in `polevals`), published in pole_gran[2 f + {0 north, 1 south}] */
    int pole_blocks; unsigned pole_epoch;
    int special_last;                 /* 1: special rows at the end of the field's work order (single-field launch) */
    int by_lo, by_cnt;                /* by_cnt > 0 (with special_last = 1, one field): the launch covers rows by_lo .. by_lo + by_cnt - 1 of the
                                         work order [segments | special rows] only (host-pointer ABI: row ranges as the source arrives) */
    int pole_timeout; int *err_word;  /* set by the launcher / the kernel: the bounded wait for the pole values gave up (polar rows := NaN, *err_word := 1) */
    unsigned long long *pole_gran;    /* [pole_blocks] {launch epoch << 32 | REAL bits of the pole value}: ONE 64-bit word per value, written and polled with relaxed agent-scope
                                         atomics: flag and value cannot be seen apart, whatever the memory model allows for two separate words */
    float pole_now[2]; int pole_inline;   /* set inside the kernel: the two values a special-row block took from its granules */
    int x_nbx; size_t x_lds_bytes;    /* set by the launcher: column strips, dynamic LDS bytes */
    /* fused compact_float min/max (k_sepx<.., STATS>): every thread block writes {min key, max key, 0} of the values it
     * stored to stat_partials[field * stat_stride + 3 * (work item in field)]; NULL: not requested */
    unsigned *stat_partials; size_t stat_stride;
    /* what a launch leaves behind (k_sepx only): 0 float fields in zout; 2 nothing but the min/max partials (compact_float's
     * first pass over values that are never stored: the cfg5 pipeline); 3 compact_float's 16-bit tokens of the values,
     * two per 32-bit word, first in the high half (zout is then the token array, batch_out_stride in WORDS; ni_dst even),
     * quantised with the {minF, mulFactor} of quant_params[field] (packhip_cf_params, quant_stride bytes apart) */
    int out_mode; const void *quant_params; size_t quant_stride;
    size_t batch_in_stride, batch_out_stride;   /* floats between consecutive fields */
    const ezhip_xstep *x_first, *x_cont;   /* [x_nvb] staging step of a row-block when it starts a segment / continues one */
    const ezhip_xrow *x_rows;         /* [x_nvb * x_rows_per_step] row records */
    /* ---- exact extrema of the interpolated field WITHOUT interpolating it (k_bb_*, the cfg5 pipeline's pass A): every value is
     * sum(w z) over its stencil = one WINDOW of bb_ntap x bb_ntap source points, so it lies within [wmin - bb_a range - bb_s zabs,
     * wmax + bb_a range + bb_s zabs] of that window (range = wmax - wmin, zabs = max |z|; bb_a = (max sum |wx| * max sum |wy| - 1) / 2 and
     * bb_s bounds |sum w - 1| and the rounding of the REAL*8 evaluation).  Only windows whose bound reaches the best GUARANTEED value can
     * hold the extremum; their target points (CSR lists by the window's first source column / row) are evaluated exactly. */
    int bb_ok, bb_ntap;
    double bb_a, bb_s;
    const unsigned char *bb_colhas, *bb_rowhas;      /* [ni_src], [nj_src]: some target column / main row has its first tap there */
    const int *bb_colstart, *bb_collist;             /* [ni_src + 1], [ni_dst] */
    const int *bb_rowstart, *bb_rowlist;             /* [nj_src + 1], [main rows] */
    const float *bb_colk, *bb_rowk;                  /* [ni_src], [nj_src]: max sum |wx| of the target columns / sum |wy| of the rows of the window (rounded up); a window's a = (colk rowk - 1) / 2: the
                                                        rows that EXTRApolate towards a pole without the polar correction carry weights far beyond the interior's 1.25 */
    /* ---- interpolate AND encode in one launch (k_sepx_enc, the cfg5 pipeline's passes B + E): a thread block owns 256 target columns (one context
     * column + 85 tiles of 3) x 16 target rows (one context row + 5 tile rows) = five chunks of the armn_compress stream; strips advance by 255
     * columns, row groups by 15 rows.  e_ok: the geometry exists (every strip staged from consecutive taps, no fill columns / rows) */
    int e_ok, e_nstrips, e_nrg, e_tr, e_prows, e_wstride;
    const int *e_blk_base, *e_blk_w;  /* [e_nstrips] staged source columns of a strip */
    const ezhip_xstep *e_step;        /* [e_nrg] the source rows of a row group (all new: no ring reuse between thread blocks) */
    const ezhip_xrow *e_rows;         /* [e_nrg][16] row records (o_off unused) */
    const int *e_special;             /* [e_nrg][16] index into special[] of a row that is not a main row, -1: main row (or past the last row) */
} ezhip_sep_plan;

/* what k_sepx_enc needs beside the plan: the streams, the look-back storage (zeroed), compact_float's parameters per field */
typedef struct {
    unsigned *z; size_t z_stride, z_cap;             /* stream of field f at z + f * z_stride, capacity z_cap words */
    const float *zin; size_t in_stride; int nfields;
    const float *poles;                              /* [2 nfields] pole values (plans with need_poles) */
    const void *quant_params; size_t quant_stride;   /* packhip_cf_params of field f: {minF, mulFactor} first */
    int nbits, container, ntx, nty, nchunks;         /* tiles per tile row, tile rows, chunks per field (nty * e_nstrips) */
    unsigned long long *status, *tail;               /* [nfields][nchunks] */
    unsigned *ctl;                                   /* [0] abort flag */
    unsigned short *ptok; size_t ptok_stride;        /* [nfields][ni + nj - 1] tokens of the stream prefix: row 0, then column 0 from row 1 on */
    unsigned *head;                                  /* [nfields] first image word of chunk 0: the bits of the stream word it shares with the prefix */
    int *zlng;                                       /* [nfields] */
    int debug;
} ezhip_sepenc_args;
/* layout of the launch's device scratch (zeroed up to off_head by the caller): granules, abort flag, chunk 0's first words, prefix tokens */
typedef struct { size_t off_status, off_tail, off_ctl, off_head, off_ptok, ptok_stride, zero_bytes, total; int ntx, nty, nstrips, nchunks; } ezhip_sepenc_layout;
static inline ezhip_sepenc_layout ezhip_sepenc_layout_of(int ni, int nj, int nfields)
{
    ezhip_sepenc_layout L;
    L.ntx = (ni - 1 + 2) / 3; L.nty = (nj - 1 + 2) / 3; L.nstrips = (L.ntx + 84) / 85; L.nchunks = L.nty * L.nstrips;
    L.off_status = 0; L.off_tail = 8 * (size_t)L.nchunks * (size_t)nfields; L.off_ctl = 2 * L.off_tail;
    L.off_head = L.off_ctl + 64; L.zero_bytes = L.off_head;
    L.off_ptok = (L.off_head + 4 * (size_t)nfields + 63) & ~(size_t)63;
    L.ptok_stride = ((size_t)ni + (size_t)nj - 1 + 7) & ~(size_t)7;
    L.total = L.off_ptok + 2 * L.ptok_stride * (size_t)nfields + 64;
    return L;
}
size_t ezhip_sepenc_lds_bytes(const ezhip_sep_plan *plan);
int ezhip_interp_sep_enc(const ezhip_sep_plan *plan, const ezhip_sepenc_args *args);

int ezhip_interp_sep(const ezhip_sep_plan *plan, float *d_zout, const float *d_zin);
/* after an ezhip_interp_pts2 launch of the calling thread that listed its special points: their number (synchronises the stream); with arrays of that many
 * elements also the points themselves (index, x, y), copied on the device.  -1 on error */
int ezhip_pts2_special_snapshot(int *d_list_out, float *d_x_out, float *d_y_out, int cap, const float *d_xs, const float *d_ys);
/* exact {min, max} of the values ezhip_interp_sep would store for nfields fields (plan->bb_ok), asynchronous: d_partials[f * stride_words + 0..2] :=
 * {min key, max key, 0} (the triple layout k_cf_header reduces); d_flags[f] := 1 when too many windows qualify (the caller then runs the
 * interpolating pass for that field); d_poles: [2 nfields] pole values from ezhip_polevals_batch (plans with need_poles).  d_work: ezhip_bb_work_bytes */
size_t ezhip_bb_work_bytes(const ezhip_sep_plan *plan, int nfields);
int ezhip_minmax_bb_special(const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                            int *d_flags, const float *d_poles, void *d_work);
int ezhip_minmax_bb(const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                    int *d_flags, const float *d_poles, void *d_work);
int ezhip_minmax_bb_stage(const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                    int *d_flags, const float *d_poles, void *d_work, int stage);
/* co-resident k_sepx thread blocks on the current device for a given dynamic LDS size (0: unknown) */
int ezhip_sepx_capacity(int degree, int rows_per_step, size_t lds_bytes);
size_t ezhip_sepx_lds_bytes(int x_tr, int rows_per_step, int x_prows, int wstride);

/*
 * Generic per-point plan ("mode B"): arbitrary located coordinates.  Restates the reference's
 * leaf kernel for the grid kind / degree / wrap, point by point, including zone handling.
 */
typedef struct {
    int degree;                       /* 0, 1, 3 */
    int irregular;                    /* 1: Z/#/G source (ax, ay, ncx, ncy tables), 0: regular */
    int ni, nj, i1, i2, j1, j2, wrap; /* source geometry, 1-based bounds as in the reference */
    const float *ax, *ay, *ncx, *ncy; /* device; NULL for regular sources */
    const float *ncx8, *ncy8;         /* the same coefficients laid out [index][8] (6 used): two 16-byte loads per point */
    const double *xrec10, *yrec10;    /* k_uvt: per source column i (row j) one 80-byte record of REAL*8 {ax(i-1), ax(i), ax(i+1), c1 .. c6, c5 + c2} (the REAL table entries converted once,
                                         not per point), staged in LDS per tile; index i - 1 (j - j1) */
    const float *xrec8, *yrec8;       /* the wind pair kernels (k_pts2_irgd3w, k_uvt): per source column i (row j) one 32-byte record of REAL {ax(i-1 .. i+2), d1 .. d4},
                                         d_k = 1 / prod_{m != k} (x_k - x_m) formed in REAL*8 and rounded once: the Lagrange weights of the stencil; index i - 1 (j - j1) */
    const void *uvt_tiles;            /* k_uvt: the grid set's tile table (int4 {i0, j0, W, H} per 32 x uvt_th tile of the target: ezhip_uvt_build), NULL: k_pts2 */
    int uvt_shape;                    /* 100 TW + TH of the table's tiles: 3232 (default), 3216, 6416, 6408 */
    int uvt_cap;                      /* staged cells a tile may need (the table was built under it) */
    int uvt_nhb;                      /* k_uvt (round 6): tiles handed back to the gathering path (seam, windows beyond the cap), listed BEHIND the table's int4 entries
                                         (unsigned[uvt_nhb] at tiles + ntiles): the launch's first uvt_nhb blocks take them, the blocks in table order skip them.  0: table order only */
    const void *uvt_streams;          /* NULL, or the set's x, y and packed rotation once more in tile order: 12 bytes per point (ezhip_uvt_pack_streams) */
    int uvt_debug;                    /* development knock-outs (develop build only) */
    int uvt_read2;                    /* development: k_uvt reads its cells with the compiler's ds_read2_b64 pairs instead of single ds_read_b64 (same results) */
    /* c_ezuvint_batch_dev: npairs > 1 wind pairs of one grid set in ONE staged-tile launch -- pair f's components at d_in + f * pair_in_stride, its results at
     * d_out + f * pair_out_stride (floats), its four polar wind rows at pw_out / pole_row_* + f * pair_rows_stride; x, y and the rotation of a point are read once
     * for all pairs */
    int npairs, pair_rows_stride;
    int cspec_inline;                 /* k_uvt: the polar-wind producer blocks take the set's (few) special points along when their rows are done -- no launch behind the kernel */
    size_t pair_in_stride, pair_out_stride;
    int wind_dst_rot;                 /* with wind_M: the TARGET frame is a rotated one (decides what a REAL overflow of the chain's speed turns into) */
    int wind_M_half;                  /* wind_M holds ONE word per point: a pure rotation (a, b), c = -b, d = a, packed by rot_pack (ez_kernels.hip) */
    const void *wind_M;               /* k_pts2: the grid pair's wind matrices (ezhip_wind_matrix), applied to every point before it is stored; NULL: store the interpolated components */
    /* zone handling (0 = none: c_gdxysint semantics) */
    int zones;                        /* 0 none, 1 EZ_NO_EXTRAP (polar zones), 2 EZ_EXTRAP (DEHORS) */
    int only_special;                 /* 1: the normal points keep what the output already holds (interp_degree = average ran before): only fill / pole / listed points are written */
    int degre_extrap;                 /* used when zones == 2 */
    float ypole_n, ypole_s;
    float ay4_n[4], ay4_s[4];         /* strip latitudes (irregular cubic) */
    int pole_weighted;
    int vector_mode;
    const float *pole_row_n, *pole_row_s;
    const float *fill;                /* device scalar */
    const float *polevals;            /* device float[2] = {north, south}; computed by ezhip_polevals */
    const int *out_idx;               /* NULL, or target position of point n (Yin-Yang lists: no temporary + scatter pass) */
    int newton_literal, xcd_order, tile_shape;    /* development switches of k_pts2 (set by its launcher from the environment) */
    /* the synthetic polar wind rows of the pair (k_polar_wind's job) riding in the k_pts2 launch as two producer blocks: only the special points, handled by
     * the NEXT kernel, read them.  pw_out != NULL: out4 = [u north | u south | v north | v south] rows of ni floats */
    float *pw_out; const float *pw_plon2; float pw_xg4_n, pw_xg4_s; int pw_weighted; const float *pw_ax;
    /* scalar per-point path: pv_out != NULL: the two pole values of the field are summed by two producer blocks at the head of the k_pts launch (they were a
     * launch of their own in front of it: 16 us of a 99 us c_ezsint from a rotated 2560 x 1280 source); the points that read them -- the pole points and the
     * polar strips -- are the NEXT kernel's.  pv_nj: rows of the field the sums run over */
    float *pv_out; int pv_nj;
    /* the special points of a grid set depend on its located x, y and the zone options only: once a launch has listed them the host keeps them with the set
     * (ezhip_pts2_special_snapshot) and later launches take them from here -- k_pts2 lists nothing, the special kernel reads n, x, y side by side instead of
     * count -> list -> x, y (three dependent round trips) */
    int cspec_valid, cspec_count; const int *cspec_list; const float *cspec_x, *cspec_y;
    int tile_ni, tile_nj;             /* > 0: the points are a whole ni x nj target grid in row order: k_pts2 walks it in 32 x 8 tiles (a wave = 8 x 8 points: its stencils
                                         share a few cache lines; 64 points of one target row cross a dozen source rows of a rotated source) */
} ezhip_pts_plan;

int ezhip_interp_pts(const ezhip_pts_plan *plan, float *d_zout, const float *d_zin,
                     const float *d_x, const float *d_y, int npts);

/* masks (ez_mask.c): mode 0 = c_ezsint_mask, 1 = c_ezget_mask_zones; x, y = located coordinates of every target point */
int ezhip_mask(int *d_mask_out, const float *d_x, const float *d_y, const int *d_mask_in, int ni_in, int nj_in, int ni_out, int nj_out, int mode, int cloud_linear);
int ezhip_mask_fill_min(float *d_fld, const int *d_mask, size_t n, unsigned *d_keys2);
/* k_uvt's tile table of a wind-pair plan over the located x, y of its grid set (tile_ni x tile_nj target in row order): d_tiles receives
 * ezhip_uvt_ntiles(plan, shape) int4 entries (shape = 100 TW + TH).  stats (host, may be NULL): [0] tiles staged, [1] handed to the gathering path, [2] empty, [3] largest window (cells).  Synchronises. */
int ezhip_uvt_ntiles(const ezhip_pts_plan *plan, int shape);
/* want_list: d_tiles holds 20 bytes per tile; the indices of the handed-back tiles (stats[1] of them) are written behind the table's entries */
int ezhip_uvt_build(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_tiles, int shape, int *stats, int want_list);
size_t ezhip_uvt_stream_bytes(const ezhip_pts_plan *plan, int shape);
int ezhip_uvt_pack_streams(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_streams, int shape);
int ezhip_interp_pts_batch(const ezhip_pts_plan *plan, float *d_zout, const float *d_zin, const float *d_x, const float *d_y, int npts, int nfields, size_t in_stride, size_t out_stride);      /* k_st over a batch: -2 when the plan is not on that path */
int ezhip_st1_build(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_tiles, int *stats);      /* k_st1: the bilinear kernel's tile table (32 x 32 tiles) */
int ezhip_st_pack_streams(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_streams);      /* k_st: {x, y} of every point in tile order (32 x 32 tiles), 8 bytes x 1024 x ntiles */
int ezhip_interp_pts2(const ezhip_pts_plan *plan_u, const ezhip_pts_plan *plan_v, float *d_out_u, float *d_out_v,
                      const float *d_in_u, const float *d_in_v, const float *d_x, const float *d_y, int npts);
/* ez_xpngdag2 / ez_xpngdb2: hemispheric field (ni x nj) -> rows j1..j2 of its global expansion, mirrored rows times +-1 */
int ezhip_hemi_expand(float *d_dst, const float *d_src, int ni, int nj, int j1, int j2, int hem, int is_b, int symetrie, int yinv);   /* hem 0: copy; yinv: source rows in reverse order first (PERMUT) */
/* d_dst[d_idx[k]] = d_src[k] (the merge of the Yin and Yang point lists) */
int ezhip_scatter(float *d_dst, const float *d_src, const int *d_idx, int n);
/* pole values {north, south} of a source field -> device float[2] */
int ezhip_polevals(float *d_out2, const float *d_zin, int ni, int nj, int weighted, const float *d_ax);
int ezhip_polevals_batch(float *d_out, const float *d_zin, size_t field_stride, int nfields, int ni, int nj, int weighted, const float *d_ax);
/* min / max of a field -> device float[2]; then fill = f(min,max) on device */
int ezhip_fill_value(float *d_fill, const float *d_zin, size_t n, int degre_extrap, float valeur, int vector_mode);

/* ---- locate ------------------------------------------------------------------------------ */
typedef struct {
    int kind;                         /* 0: regular lat-lon (llll2gd), 1: irregular axes on 'L' ref (G, Z/L),
                                         2: irregular axes on rotated 'E' ref (Z/E), 3: regular rotated 'E',
                                         4: polar stereographic N / S: (lat0, lon0, dlat, dlon) = (pi, pj, d60, dgrw), lon_fix = hemisphere */
    int ni, nj;
    float lat0, lon0, dlat, dlon;     /* kind 0/3: llll2gd parameters; kind 1: reference-grid decode */
    float lonref;                     /* kind 1: 0 or -180 */
    int   lon_fix;                    /* kind 0: 1 = 'L' wrap fix (lon0, ni*dlon), 2 = "<0 -> +360" */
    float r[9];                       /* kind 2/3: rotation matrix (Fortran order) */
    const float *ax, *ay;             /* device axes (kind 1/2) */
} ezhip_locate_plan;

/* separable target: lat1d[nj_dst], lon1d[ni_dst] (device); full target: lat2d/lon2d [npts] */
int ezhip_locate(const ezhip_locate_plan *plan, float *d_x, float *d_y,
                 const float *d_lat, const float *d_lon, int ni_dst, int nj_dst, int separable);

/* libm_exact.h evaluated on the device (fn 0 sinf, 1 cosf, 2 asinf, 3 atanf, 4 atan2f(a, b)); tests compare with the host's C library */
int ezhip_libm_exact_probe(int fn, const float *d_a, const float *d_b, float *d_out, size_t n);
/* *d_flag = 1 if any located point lies outside 1 .. ni, 1 .. nj after rounding (the DEHORS zone's test), else 0 */
int ezhip_any_dehors(const float *d_x, const float *d_y, size_t n, int ni, int nj, int *d_flag);

/* ---- winds ------------------------------------------------------------------------------- */
typedef struct {
    int src_rotated;                  /* 1: source is E / Z-on-E: rotate through ri */
    float r[9], ri[9];
    int separable;                    /* target lat/lon given as 1-D arrays */
    int wd_only;                      /* 1: stop after c_gdwdfuv (c_ezwdint): uu := speed, vv := direction */
    int fast_trig;                    /* 1: REAL instead of REAL*8 sine / cosine of the rotated coordinates (development switch EZHIP_WIND_FAST_TRIG) */
    int wd_in;                        /* 1: uu / vv already hold speed / direction: only c_gduvfwd on the target (Yin-Yang merge) */
    int src_ps, dst_ps;               /* 0, or 1 = 'N' / 2 = 'S': polar-stereographic source / target (ez_llwfgdw.inc:91-140, ez_gdwfllw.inc:93-121) */
    float src_xg4, dst_xg4;           /* their dgrw */
    int dst_rotated; float r_dst[9];  /* Z-on-E TARGET: c_ezgfwfllw (ez_gfwfllw.c:38-79) with its rotation matrix r */
    const double *lon_trig, *lat_trig;   /* separable + rotated source: {cos, sin} per target column / row (ezhip_wind_trig_tables), or NULL */
    const float *lon_trigf, *lat_trigf;  /* the REAL {cos, sin} pairs of the same angles (rotation into the source frame) */
    /* Lambert '!' source / target (ez_lamb_llwfgdw.inc, ez_lamb_gdwfllw.inc): {cos, sin} of the grid's rotation angle at the point's longitude, from the host
     * (the angle comes from two projections of the reference's REAL libm chain); one pair per target column when `separable`, else per point */
    const float *src_lamb_cs, *dst_lamb_cs;
} ezhip_wind_plan;
int ezhip_wind_trig_tables(double *d_lon_trig, double *d_lat_trig, float *d_lon_trigf, float *d_lat_trigf,
                           const float *d_lat, const float *d_lon, int ni, int nj);

/* synthetic polar wind rows of a source (u,v) pair: d_out4 = [u_n, u_s, v_n, v_s], ni floats each; d_plon2 = longitudes of
 * the last and the first source row */
/* fork / join of a per-thread side stream (see ez_kernels.hip) */
int ezhip_side_begin(void);
int ezhip_side_end(void);
int ezhip_side_join(void);
int ezhip_polar_wind(float *d_out4, const float *d_uu, const float *d_vv, const float *d_plon2, int ni, int nj,
                     float xg4_n, float xg4_s, int weighted, const float *d_ax, int exact);

/* in place on (uu, vv): source-grid components -> target ('L'-like) grid components */
/* the chain of a grid pair as a 2 x 2 matrix per point (16 bytes each): built once, applied per call */
int ezhip_wind_matrix(const ezhip_wind_plan *plan, void *d_M, const float *d_lat, const float *d_lon, int ni_dst, int nj_dst, float *max_dev);   /* d_M: 24 bytes per point */
int ezhip_corrbgd(float *d_zout, int ni, int nj, int hem);
int ezhip_wind_apply(const void *d_M, int half, float *d_uu, float *d_vv, size_t npts, int dst_rotated);
/* interp_degree = average / sph_average (ez_avg.inc, ez_avg_sph.inc): bounds = [x ni_dst | row widening nj_dst | y_low nj_dst | y_high nj_dst] on the device */
int ezhip_average(float *d_zout, const float *d_zin, const float *d_bounds, int ni_dst, int nj_dst, int ni_src, int nj_src, int extension, float ylast);
int ezhip_wind_rotate(const ezhip_wind_plan *plan, float *d_uu, float *d_vv,
                      const float *d_lat, const float *d_lon, int ni_dst, int nj_dst);

#ifdef __cplusplus
}
#endif
#endif
