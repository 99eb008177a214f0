/*
 * qbhip.h -- C ABI of libqbhip.so, the MI355X (gfx950) engine for the sparse
 * Hamiltonian x vector hot path of wztzjhn/quantum_basis.
 *
 * The reference has no FFI layer; its operator boundary is the duck-typed MAT
 * template parameter of lanczos<T,MAT> / eigenvec_CG<T,MAT> / iram<T,MAT>
 * (src/qbasis.h:1065-1093), closed at link time over csr_mat<T>
 * (src/qbasis.h:976-1021).  Every entry point below replaces one reference
 * function; the citation says which.  Plain pointers and sizes only: no C++
 * types, no torch types, no exceptions cross this boundary (functions return
 * 0 or a negative QBH_E* code; qbh_last_error() gives the detail).
 *
 * Complex numbers are interleaved (re, im) doubles, layout-compatible with
 * std::complex<double> and MKL_Complex16.  All index integers are 64-bit on
 * the host side (the reference builds with -DMKL_ILP64).
 *
 * Threading: like the reference (single caller thread, src/model.cc:1177),
 * calls on one operator handle must not be concurrent.
 *
 * There is no CPU fallback: every compute entry point fails with
 * QBH_ENODEVICE when no HIP device is present.
 */
#ifndef QBHIP_H
#define QBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QBH_VERSION 601

/* error codes */
#define QBH_OK          0
#define QBH_EINVAL     -1   /* invalid argument (reference: std::invalid_argument / assert) */
#define QBH_ENODEVICE  -2   /* no HIP device / HIP runtime error at init                    */
#define QBH_EHIP       -3   /* a HIP call failed (reference: std::runtime_error from MKL)   */
#define QBH_ENOMEM     -4   /* host or device allocation failed                             */
#define QBH_ENOTHERM   -5   /* full-storage matrix failed the Hermitian check (sparse.cc:235-256, reference exits 99) */
#define QBH_ECOMM      -6   /* a communicator hook reported failure                         */
#define QBH_ENOTNORM   -7   /* Lanczos start vector not normalised (assert at lanczos.cc:166) */
#define QBH_ENOCONV    -8   /* tridiagonal QL failed to converge                            */
#define QBH_EUNSUPP    -9   /* valid in the reference but not supported here                */

typedef struct qbh_z { double re, im; } qbh_z;

/* opaque device-resident operator: one row shard of H in HBM, its stream, workspace */
typedef struct qbh_csr qbh_csr;

/* tolerances reused verbatim from src/miscellaneous.cc:44-47 */
#define QBH_LANCZOS_PRECISION 2e-12
#define QBH_SPARSE_PRECISION  1e-14

/* ---------------------------------------------------------------- misc --- */
int         qbh_version(void);
int         qbh_device_count(void);                 /* 0 when there is no usable GPU */
const char *qbh_strerror(int code);
const char *qbh_last_error(void);                   /* thread-local detail of the last failure */

/* ------------------------------------------------------------- options --- */
#define QBH_KERNEL_AUTO    0   /* coded values: QBH_KERNEL_ROWS; complex128 values: QBH_KERNEL_WAVE */
#define QBH_KERNEL_STREAM  1   /* row blocks, val*x products through LDS, TPR lanes per row   */
#define QBH_KERNEL_VECTOR  2   /* sub-wavefront per row, shuffle reduction, no LDS            */
#define QBH_KERNEL_ROWS    3   /* row blocks staged in LDS, lanes mapped to rows (default)    */
#define QBH_KERNEL_WAVE    5   /* one wavefront per block of whole rows (<= 512 nonzeros), no workgroup barrier:
                                  complex128 values only (coded operators use the row kernel)            */
#define QBH_KERNEL_MATRIX_FREE 4 /* reported by qbh_csr_get_info for qbh_mf_hubbard operators   */

/* qbh_opts.basis_kind */
#define QBH_BASIS_NONE          0   /* an index is an index (default) */
#define QBH_BASIS_REF_FERMION2  1   /* the reference's order of a two-species fermion basis (see qbh_opts.basis_kind) */
#define QBH_BASIS_DETECT       -1   /* qbh_csr_set_basis only: run the search of qbh_opts.basis_detect on an existing operator */
#define QBH_BASIS_SPIN_SECTOR   2   /* spin-1/2 (one species), n_dn particles on n_sites sites, index = rank of the bit pattern in ascending
                                       order (qbh_gen_heisenberg; n_up unused): the sites are cut into a low and a high half
                                       (qbh_opts.n_up = number of low sites, 0 = n_sites / 2) and the operator is held class-major
                                       internally -- class = particle number of the high half -- so that bonds inside a half get the
                                       Kronecker treatment and only the bonds across the cut stay unstructured (kagome-30: 17-21 %) */
#define QBH_BASIS_SECTOR_ORBIT  3   /* reported by qbh_csr_info.basis_internal only: a matrix-free momentum-sector operator whose rows
                                       are held orbit by orbit of the up patterns (qbh_opts.sector_orbit); not a value of basis_kind */

typedef struct qbh_opts {
    int     device;          /* HIP ordinal; -1 = current device                                  */
    void   *stream;          /* hipStream_t to enqueue on; NULL = library-owned stream             */
    int     spmv_kernel;     /* QBH_KERNEL_*                                                       */
    int     nnz_per_block;   /* nonzeros staged per workgroup (1024/2048/4096/8192); 0 = auto      */
    int     xcd_swizzle;     /* workgroup->row-block map: 0 interleaved, 1 one contiguous eighth per
                                XCD, 2 (default) chunked: neighbouring chunks on the 8 XCDs          */
    int     value_dict;      /* 0 = store complex128 values; 1 = dictionary-code them (exact, lossless):
                                1-byte codes when <= 256 distinct values exist, 2-byte codes when
                                <= 65536 (row kernel only); 2 = 1-byte codes only                  */
    int     profile;         /* 1: bracket every SpMV launch with HIP events (qbh_get_stats)       */
    int     check_hermitian; /* 1 (default): full-storage host input is checked like sparse.cc:235 */
    int     real_fast_path;  /* 1 (default): a real operator applied to vectors with exactly zero imaginary
                                parts gathers / exchanges 8-byte real parts (bit-identical results);
                                0: always the complex128 arithmetic of the reference (north-star format)   */
    int     kron_split;      /* 1 (default): a complex128 operator on a product basis (index = major * S + minor, every entry
                                changes one of the two: the two-species Hubbard family in species-major order) is stored as
                                H_near + H_far, the far part band-major over the minor index -- the handle's arrays are
                                RE-ORDERED IN PLACE (same values in the same order, columns as int32 or -- kron_cols16 -- 2 bytes each; no second copy;
                                qbh_csr_download merges the parts back).  The structure is verified on the device, the choice
                                is structural (never timed).  Row shards made of whole major indices (row_offset and the
                                row count multiples of S) split the same way.  1 leaves operators below 1e8 nonzeros alone
                                (three launches cost more than they save there); 2: whatever has the structure; 0: never.
                                The library's DEFAULT format (value_dict + real_fast_path: 1-byte value codes applied to
                                packed-double vectors) follows the same rule with a split of its own: a second, re-ordered
                                copy of the coded operator (3 B per nonzero beside the 5 B per nonzero of the coded CSR, which
                                stays for complex vectors, shards and qbh_csr_download) whose near pass gathers from the block
                                of x held in LDS -- taken when S doubles fit one workgroup's LDS (S <= 20480) and the major
                                count fits 16 bits; used by the all-real solves only (qbh_csr_info.kron_minor / kron_sliced).
                                Where the far entries of a row do not depend on its minor index and the near ones, the diagonal
                                apart, not on its major index (H = T (x) 1 + 1 (x) T' + D: the two-species models), that is
                                recognised on the device and the parts are kept once (T, T', one diagonal code per row)        */
    int64_t kron_minor;      /* S for an operator created from host / device arrays (0: unknown -> no split); the generators
                                announce their own; with basis_kind set it is derived from the hint                   */
    int     deterministic;   /* 1: nothing about the operator is decided by a clock and nothing in an SpMV depends on the
                                order in which wavefronts finish: no kernel timing at creation (QBH_KERNEL_AUTO keeps its
                                default form), static walks instead of the per-XCD work counters.  Two runs on any two
                                MI355X then give bit-identical vectors, a_j / b_j and step counts.  0 (default): see
                                INTEGRATION.md section 4 for what is and is not reproducible                            */
    int     basis_kind;      /* QBH_BASIS_*: what the row / column index of host arrays means.  QBH_BASIS_REF_FERMION2: the
                                reference's own order for a two-species fermion model (mbasis_elem bit order, Lin tables of
                                src/basis.cc:1144-1190, operators ordered by site, src/basis.cc:2650-2664) on n_sites sites
                                with n_up + n_dn particles.  The library then keeps the operator INTERNALLY in species-major
                                order (index = up * C(n_sites, n_dn) + down, so that the Kronecker split applies) and
                                permutes / sign-flips vectors at the seams that copy them anyway (qbh_multmv/2, qbh_lanczos,
                                qbh_eigenvec_cg, qbh_iram host vectors): callers see the reference's order throughout.  A hint
                                that does not describe the matrix is detected (the permuted operator must have the product
                                structure) and the operator is then kept exactly as given                              */
    int     n_sites, n_up, n_dn;
    int     kron_cols16;     /* 1 (default): the parts of a split two-species operator keep 2-byte columns where they fit -- near:
                                relative to the major index of the wave block's first row (S <= 32768); far (whole operator, sliced):
                                the target major index (major count <= 32768); 16 instead of 20 B of stream per nonzero, the same
                                values in the same order: results are bit-identical to the int32 form.  qbh_csr_download re-derives
                                the int32 columns.  0: int32 columns (SURVEY 8(d)'s format to the byte)                     */
    /* ---- which FORM of an equivalent computation runs (summation order may differ between forms; every form is tested
     * against the oracle).  These were environment switches up to ABI 400; the library reads no environment variable that
     * changes a result any more (QBH_DEBUG holds measurement / tracing knobs only, QBH_HOST_THREADS the host thread count,
     * QBH_RCCL_LIB the communication library).  qbh_opts_default() sets the values in brackets. ---- */
    int     kron_sliced;     /* [1] far part of the split interleaved in groups of 8 rows where that costs < 1/8 padding; 0: plain
                                rows in tiled order; 2: sliced whenever a group fits the wave tile (padding accepted)          */
    int     kron_band;       /* [0] band width of the tiling: 0 = one 128-byte line per major index, narrower while a band of x
                                exceeds an XCD's L2; 2 / 4 / 8 / 16 force it                                                  */
    int     kron_cross_in_near; /* [1] cut single-species sector (QBH_BASIS_SPIN_SECTOR): the entries across the cut stay in the
                                near part (two passes); 0: a third pass of their own                                       */
    int     kron_coded;      /* [-1] split of the DEFAULT (coded, real) format: -1 follows kron_split; 0 never; 1 the two-part row
                                kernel form (measured slower, kept for comparison); 2 the sliced form (qbh_kronc.hip)        */
    int     kron_uniform;    /* [7] sliced coded split: bit 0 recognise a far part T (x) 1, bit 1 a near part 1 (x) T' + D and keep
                                T / T' once; bit 2: when BOTH are recognised apply the operator with the row-staged table kernel
                                (T, T' as hop tables, one diagonal code per row: no tiled copy, no far sums) instead of the two
                                sliced passes (qbh_csr_info.kron_table_kernel); 0: every group stored                       */
    int     gather_parts;    /* [0] band ranges the gather of x travels in under a communicator with part hooks: 0 = 4 when
                                there are peers, else 1; 1..8 force it.  EVERY rank must pass the same value                  */
    int     wave_walk;       /* [-1] walk of the wave kernels over their blocks: -1 = the ordered per-XCD work counters (static
                                chunked walk when deterministic); 0 interleaved, 1 contiguous eighths, 2 chunked, 3 counters   */
    int     tile_fold;       /* [1] inside the solvers the pass that produces the next x also writes its tiled copy; 0: every
                                SpMV makes the copy itself (what a caller of qbh_spmv_dev always gets)                       */
    int     autotune;        /* [1] operators without product structure above 1e7 nonzeros: row kernel vs wave kernel timed at
                                creation (never when deterministic); 0: the wave kernel                                    */
    int     shard_split;     /* [1] a plain row shard under a communicator is split into locally-owned and remote columns so
                                that the local part overlaps the all-gather; 0: one launch after the gather                  */
    int     real_forms;      /* [7] with real_fast_path, for real operators and vectors: bit 0 only real parts travel in the
                                all-gather, bit 1 the row kernel gathers 8-byte real parts, bit 2 the solvers keep their
                                vectors as packed doubles                                                                  */
    int     basis_detect;    /* [1] arrays that arrive without basis_kind and without kron_minor (the reference's
                                csr_mat(lil_mat&) carries no options), whole operator, large enough for the split
                                (kron_split = 1: >= 1e8 nonzeros; 2: any): every (n_sites, n_up, n_dn) with C(n_sites, n_up) *
                                C(n_sites, n_dn) = dim is tried as QBH_BASIS_REF_FERMION2 -- one pass over the columns per
                                candidate, BEFORE anything is permuted -- and the first under which the operator has the
                                product structure is taken (qbh_csr_info.basis_internal / basis_detected / basis_n_*).  A
                                matrix of a colliding dimension without the structure stays exactly as given.  0: never     */
    int     sector_orbit;    /* [1] qbh_mf_hubbard_repr: the rows of a down block (one per up pattern) are held ORBIT BY ORBIT of
                                the translation group, so that a translated pattern lies inside the same 8 x n_trans bytes:
                                every down hop of a block then reads its target block front to back once, instead of
                                gathering through a per-translation rank table.  The handle keeps its device vectors in that
                                order (qbh_csr_info.basis_internal = QBH_BASIS_SECTOR_ORBIT; host vectors are translated at the
                                seams as for basis_kind, device vectors by qbh_vec_to_internal / _from_internal).  0: rows in
                                ascending pattern order, the form of ABI <= 500                                            */
    int     lanczos_pipeline; /* [1] qbh_lanczos(_dev), one GPU, complex vectors, purpose sr_val0 / dnmcs: the three-term step
                                keeps a_{m-1}, b_m and 1/b_m on the device and step m + 1 is enqueued BEFORE step m's two scalars
                                have been read back; the Ritz test (src/lanczos.cc:229-245) runs one step behind, and on
                                convergence or breakdown the one speculative step is discarded (it wrote to a third vector the
                                handle keeps), so m, a[], b[], the two vectors returned and the stop step are exactly those of
                                the unpipelined loop.  0: one host synchronisation per step (the form of ABI <= 501)       */
    int     real_wire;       /* [1] row shards of an operator split in place (kron_split) under a communicator: when the operator's
                                values and the vectors of a solve are exactly real (the all-reduced sum of |Im|^2 of the start
                                vectors is 0), the tiled blocks travel as 8-byte real parts (qbh_comm.d_xfull_r) and are expanded
                                when they are moved to their place in the tiled x -- half the bytes on the links, the same
                                numbers in the SpMV.  Independent of real_fast_path (which selects the real FORMS of the unsplit
                                kernels).  0: 16-byte elements                                                             */
    int     sparse_gather;   /* [1] split shards under a communicator that offers qbh_comm.exchange_v (the native one does): every rank
                                learns at attach time which of its major indices each peer's far / cross entries read and sends each
                                peer only those (packed per destination, in the same band ranges as the gather in parts); the
                                receiver moves them to their place in its tiled x.  C3, 8 ranks, generator's order: 42-68 % of the
                                all-gather's bytes.  0: every rank's whole block travels to everybody                           */
    int     major_partition; /* [0] qbh_gen_hubbard: P > 1 orders the up configurations (the MAJOR index of the product basis) so that P
                                consecutive blocks of them -- the row shards of dist.kron_row_cuts / (q * NU) / P -- are the parts of
                                a recursive spectral bisection of the up-hop graph instead of ranges of ascending bit patterns: a
                                rank's far part then reads far fewer of its peers' major indices (C3, 8 ranks: 0.30 instead of
                                0.42-0.68 of the all-gather; 4 ranks 0.45 instead of 0.93; tools/needed_columns.py), which is what the
                                personalised exchange carries.  Eigenvalues do not depend on the order; vectors of such an operator
                                are in ITS order (qbh_csr_info.major_partition, qbh_csr_major_order gives the map), except that
                                qbh_vec_randomize still draws element (u, d) from position (generator's u) * S + d of the stream, so
                                the start vector is the same physical vector for every P.  Deterministic (every rank computes the
                                same order from the same tables).  0: ascending bit patterns                                    */
    int     sector_cut;      /* [0] qbh_gen_heisenberg, whole operator, complex128 values (value_dict = 0, real_fast_path = 0), large enough
                                for the split (kron_split = 1: >= 1e8 nonzeros; 2: any) and no basis_kind named: the sites are cut into
                                h LOW sites and the rest, the operator is held class-major (class = particles among the high sites) and
                                split into near (bonds inside the low sites) / far (inside the high sites) / cross parts -- the form
                                of QBH_BASIS_SPIN_SECTOR (qbh_csr_info.basis_internal = 2; host vectors are translated at the seams,
                                device vectors by qbh_vec_to_internal / _from_internal).  0: the library picks h -- the feasible cut
                                (both parts at least max(4, n_sites / 4) sites, every class's band of x inside an XCD's L2, every
                                near window inside it too) with the fewest bonds
                                across it, the one nearest n_sites / 2 among equals; nothing feasible: the operator stays as generated.
                                > 0: that h.  -1: never (the rows stay in ascending pattern order, the form of ABI <= 501)      */
    int     comm_reserve;    /* [0] split shards under a communicator of more than one rank (ABI 601).  The two passes of a split shard are
                                PERSISTENT launches that fill every CU with as many wavefronts as their registers allow (2 x 208 /
                                3 x 168 of the 512 per SIMD lane); RCCL's own send / receive kernel (rcclGenericKernel, gfx950: 256
                                threads, 261-280 registers per lane, 19.7 KB LDS) cannot be placed beside them, and a kernel that is
                                not resident moves nothing: the exchange would start when the near pass ends.  So the passes leave
                                this many workgroups OUT of their grids -- one CU per workgroup runs one wavefront per SIMD fewer, which
                                is where a workgroup of RCCL fits -- and the far pass runs with at most 2 workgroups per CU.
                                Measured with a kernel of RCCL's footprint in place of the exchange (tests/stub_rccl solo mode,
                                profiles/r6_bench/solo_rank/occupancy_SUMMARY.txt): C3, 50 GB/s links, without: +17 % / +28 % / +20 %
                                per Lanczos step at 8 / 4 / 2 ranks; with 32-64 workgroups left out: the step of the model that
                                needs no CUs, at a cost of 0-2 %.  0: 64 workgroups (RCCL uses at most that many channels);
                                > 0: that many (rounded down to a multiple of 8); -1: none                                      */
} qbh_opts;

void qbh_opts_default(qbh_opts *o);
/* Process-wide defaults: what qbh_opts_default() returns and what a NULL `opts` argument means from now on (NULL restores the
 * built-in ones).  For a host whose constructor call cannot carry options -- the reference's csr_mat(lil_mat&) -- e.g. to name
 * the basis of the matrices it is about to hand over (basis_kind, n_sites, n_up, n_dn). */
void qbh_opts_set_default(const qbh_opts *o);

/* ------------------------------------------------------------ operator --- */
/* Replaces csr_mat<T>::csr_mat(lil_mat<T>&) + create_handle (src/sparse.cc:202-260).
 * Host CSR exactly as the reference holds it (src/qbasis.h:979-985): zero-based,
 * ia[dim+1], ja[nnz] int64, val[nnz] complex128; sym_upper != 0 means only col >= row is
 * stored (reference default).  The arrays are COPIED (and for sym_upper expanded to both
 * triangles, columns narrowed to int32) into HBM; the caller keeps ownership and may free
 * or mutate them afterwards (call_feast does, src/lanczos.cc:626-629). */
int qbh_csr_create(qbh_csr **out, int64_t dim, int64_t nnz, int sym_upper,
                   const int64_t *ia, const int64_t *ja, const qbh_z *val,
                   const qbh_opts *opts);

/* The same constructor for ONE row block of the operator (the CSR that the unchanged host code
 * assembles, src/model.cc:619-685, sharded over the GPUs of a node).  Every rank passes the SAME host arrays and its
 * own [row_begin, row_end); the handle holds those rows of the FULL operator (for Hermitian-upper input that includes
 * the mirrored entries whose upper-triangle twin lives in an earlier row) with global columns.  The host arrays are
 * streamed through pinned staging buffers; no second host copy is made. */
int qbh_csr_create_rows(qbh_csr **out, int64_t dim, int64_t nnz, int sym_upper,
                        const int64_t *ia, const int64_t *ja, const qbh_z *val,
                        int64_t row_begin, int64_t row_end, const qbh_opts *opts);

/* Row cuts for nranks row blocks balanced by the nonzeros of the FULL operator (momentum-sector matrices hold
 * decoupled one-entry rows, src/model.cc:737-740, so uniform row blocks are not work-balanced): cuts[0] = 0 <=
 * cuts[1] <= ... <= cuts[nranks] = dim.  Host only. */
int qbh_balanced_row_cuts(int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia, const int64_t *ja,
                          int nranks, int64_t *cuts /* [nranks+1] */);

/* Adopt a row shard that already lives in HBM (device-side generator, multi-GPU row
 * blocks).  Rows [row_offset, row_offset+nrows) of a global ncols x ncols operator in
 * FULL storage; d_ia[nrows+1] is local (d_ia[0] == 0); d_ja holds GLOBAL columns.
 * take_ownership != 0: the arrays were allocated with hipMalloc and belong to the library from the
 * moment of the call: they are freed by qbh_csr_destroy, or before returning when the call fails. */
int qbh_csr_create_device(qbh_csr **out, int64_t nrows, int64_t ncols, int64_t row_offset,
                          int64_t nnz, int64_t *d_ia, int32_t *d_ja, qbh_z *d_val,
                          int take_ownership, const qbh_opts *opts);

/* Replaces csr_mat<T>::destroy / ~csr_mat (src/sparse.cc:150-186).  NULL is a no-op. */
void qbh_csr_destroy(qbh_csr *A);

typedef struct qbh_csr_info {
    int64_t nrows, ncols, row_offset, nnz;   /* nnz as applied (full storage)                 */
    int64_t n_blocks;                        /* workgroups per SpMV launch                     */
    int64_t bytes_matrix;                    /* HBM bytes held by the matrix arrays            */
    int64_t bytes_algorithmic;               /* nnz*20 + (nrows+1)*8 + nrows*16 + nrows*16     */
    int     kernel;                          /* QBH_KERNEL_* actually selected                 */
    int     value_dict;                      /* number of dictionary entries, 0 = not coded    */
    int     device;
    void   *stream;
    double  create_ms;                       /* wall ms of qbh_csr_create(_rows): validation + upload + expansion +
                                                geometry (0 for device-built operators)                          */
    int64_t create_bytes_in;                 /* host bytes that call read: nnz*24 + (dim+1)*8                     */
    int64_t kron_minor;                      /* Kronecker split active (qbh_opts.kron_split): minor size S, else 0  */
    int64_t kron_far_nnz;                    /* nonzeros of the far part (band-major)                               */
    int     kron_band;                       /* band width of the tiling                                            */
    int     kron_sliced;                     /* 1: far part interleaved inside groups of 8 rows (one x line per 8 lanes) */
    int     kron_inplace;                    /* 1: the arrays hold [near | far] instead of a CSR (no second copy)           */
    int     tuned;                           /* -1: the SpMV form was not timed at creation; 0 / 1: it was, the row / wave kernel won */
    double  tune_ms_rows, tune_ms_wave;      /* the two times of that comparison (0 when not timed)                         */
    int     basis_internal;                  /* QBH_BASIS_*: != 0 when the operator is held in another order than the caller's */
    int     kron_classes;                    /* classes of the product structure: 1 two-species operators, > 1 a cut single-species sector */
    int64_t kron_cross_nnz;                  /* nonzeros of the third (unstructured) part                                       */
    int     gather_parts;                    /* communicator attached: band ranges the gather of x travels in (1 = one gather)   */
    int     kron_cols16;                     /* bit 0: the near part holds 2-byte columns, bit 1: the far part (qbh_opts.kron_cols16)   */
    int     basis_detected;                  /* 1: basis_internal was found by the library itself (qbh_opts.basis_detect)               */
    int     basis_n_sites, basis_n_up, basis_n_dn;   /* the basis named by the caller or found (0 when basis_internal == 0)          */
    double  basis_detect_ms;                 /* wall ms the search took (inside create_ms for host arrays), whatever it found          */
    int     kron_table_kernel;               /* 1: the coded split was recognised as T (x) 1 + 1 (x) T' + D and the all-real SpMV runs the table kernel */
    int     wire_element_bytes;              /* communicator attached: bytes per element of x the LAST gather put on the links -- 16 (complex128) or 8
                                                (real parts only: qbh_opts.real_wire on split shards, the real fast path on plain ones); 0 before the first */
    int     major_partition;                 /* > 1: the major indices are in the partition order of qbh_opts.major_partition (that many parts) */
    int     gather_sparse;                   /* 1: the exchange is personalised (qbh_opts.sparse_gather): only the needed major indices travel */
    double  gather_needed_frac;              /* split shard under a communicator: the share of its peers' major indices that its far / cross entries read
                                                (only those are moved into the tiled x; a sparse exchange would carry this share of the all-gather); else 1 */
} qbh_csr_info;
int qbh_csr_get_info(const qbh_csr *A, qbh_csr_info *info);

/* What qbh_opts.basis_kind says at creation, for an operator that already exists (a plain unsharded complex128 CSR: created
 * with value_dict = 0, kron_split = 0): the operator is re-ordered internally (needs room for a second copy of the matrix
 * while it runs) and split when the order described has the product structure; returns QBH_OK and changes nothing when the
 * hint does not describe the matrix (qbh_csr_info.basis_internal tells).  Device vectors made before the call are in the old order. */
int qbh_csr_set_basis(qbh_csr *A, int basis_kind, int n_sites, int n_up, int n_dn);

/* Level-1 seam, host vectors.  Replace csr_mat<T>::MultMv (y = H x, src/sparse.cc:291-297)
 * and csr_mat<T>::MultMv2 (y += H x, src/sparse.cc:262-289).  x and y are host pointers of
 * length dim and may be interior pointers (ARPACK's workd, src/lanczos.cc:476); they are
 * staged through HBM inside the call. */
int qbh_multmv(const qbh_csr *A, const qbh_z *x_host, qbh_z *y_host);
int qbh_multmv2(const qbh_csr *A, const qbh_z *x_host, qbh_z *y_host);

/* ------------------------------------------------------ device vectors --- */
/* std::vector<T> of the callers (src/model.cc:1158-1164), but in HBM. */
int qbh_vec_alloc(qbh_z **d_out, int64_t n);
int qbh_vec_free(qbh_z *d);
int qbh_vec_upload(const qbh_csr *A, qbh_z *d_dst, const qbh_z *h_src, int64_t n);
int qbh_vec_download(const qbh_csr *A, qbh_z *h_dst, const qbh_z *d_src, int64_t n);
int qbh_vec_zero(const qbh_csr *A, qbh_z *d, int64_t n);
/* Device vectors of an operator that is held in another order than the caller's (qbh_csr_info.basis_internal != 0) are in the
 * INTERNAL order.  These two copy one vector of A's row count between the orders on the device (d_dst != d_src); for an
 * operator held in the caller's order they are plain copies. */
int qbh_vec_to_internal(const qbh_csr *A, qbh_z *d_dst_internal, const qbh_z *d_src_caller);
int qbh_vec_from_internal(const qbh_csr *A, qbh_z *d_dst_caller, const qbh_z *d_src_internal);
/* Replaces vec_randomize (src/miscellaneous.cc:371-386): element j (GLOBAL index
 * row_offset + j) gets minstd_rand0 draw number j+1 of `seed`, times 1/2147483647, minus
 * 0.5, imaginary part 0; then the vector is scaled by 1/||x|| (global norm under a
 * communicator).  seed == 0 gives the constant vector 1/sqrt(n_global). */
int qbh_vec_randomize(const qbh_csr *A, qbh_z *d_x, uint32_t seed);

/* ------------------------------------------- device building blocks ------ */
/* y <- alpha*(H x) + beta*y + gamma*x_local, all vectors device-resident, shard-local
 * length nrows.  Under a communicator x is gathered first (hook allgather_x).
 * If red != NULL it receives {Re<x,y>, Im<x,y>, ||y||^2} of the NEW y (global sums).
 * MultMv is (1,0,0), MultMv2 is (1,1,0), the Lanczos step is (1/b', -b, 0), the CG
 * step (src/lanczos.cc:320-322) is (1, 0, eps-E0). */
int qbh_spmv_dev(const qbh_csr *A, const qbh_z *d_x, qbh_z *d_y,
                 double alpha, double beta, double gamma, double *red /* [3] or NULL */);
/* conj(x).y (src/lanczos.cc:47-53) -> res[2], global */
int qbh_dotc_dev(const qbh_csr *A, const qbh_z *d_x, const qbh_z *d_y, double *res);
/* y += alpha*x then ||y||^2 (cblas_zaxpy + cblas_dznrm2 fused; src/lanczos.cc:206-208) */
int qbh_axpy_norm_dev(const qbh_csr *A, qbh_z alpha, const qbh_z *d_x, qbh_z *d_y, double *nrm2_sq);
/* x *= a (src/lanczos.cc:214) */
int qbh_scal_dev(const qbh_csr *A, double a, qbh_z *d_x);
/* ||x|| (src/lanczos.cc:28-33), global */
int qbh_nrm2_dev(const qbh_csr *A, const qbh_z *d_x, double *nrm);

/* ---------------------------------------------------------- solvers ------ */
/* one row of log_Lanczos_<purpose>.txt (src/lanczos.cc:102-128) */
typedef struct qbh_lanczos_row {
    int64_t k;
    double  ritz[4];
    double  a_km1, b_k, accuracy, accu_E0, accu_E1;
} qbh_lanczos_row;

typedef struct qbh_solver_info {
    qbh_lanczos_row *log;      /* in: caller buffer or NULL; out: filled rows                     */
    int64_t          log_cap;  /* in: capacity of log (rows); rows beyond it are dropped          */
    int64_t          log_len;  /* out                                                             */
    int64_t          n_matvec; /* out: SpMV launches                                              */
    int64_t          n_reorth; /* out: re-orthogonalisations against phi0 (sr_val1)               */
    double           ms_total; /* out: wall ms inside the call                                    */
    double           ms_spmv;  /* out: sum of HIP-event SpMV kernel ms (opts.profile)             */
    double          *cg_resid; /* in: NULL or room for maxit+1 doubles (log_CG.txt residuals)     */
    /* Lanczos convergence bookkeeping (what the reference checkpoints in lczs_mlns.dat,
     * src/ckpt.cc:252-257): always written on exit; read on entry when resume != 0.          */
    int64_t          resume;
    int64_t          cnt_accuE0;
    double           accuracy, theta0_prev, theta1_prev;
} qbh_solver_info;

/* Replaces lanczos<T,MAT> for MAT = csr_mat (src/lanczos.cc:134-266), purposes "sr_val0",
 * "sr_val1", "dnmcs"; contract of src/qbasis.h:1028-1067: v holds v[k-1], v[k] at
 * (j%2)*n (phi0 at 2*n for sr_val1), hessenberg has leading dimension maxit with b[j] at
 * [j] and a[j] at [maxit+j]; same stop rule and tolerances (src/lanczos.cc:216,230-245).
 * n is the shard-local length (== dim without a communicator).  The _dev form keeps v in
 * HBM for the whole call (level-2 seam); the host form uploads v on entry and downloads
 * it on exit.  info may be NULL. */
int qbh_lanczos(const qbh_csr *A, int64_t k, int64_t np, int64_t maxit, int64_t *m,
                qbh_z *v_host, double *hessenberg, const char *purpose, qbh_solver_info *info);
int qbh_lanczos_dev(const qbh_csr *A, int64_t k, int64_t np, int64_t maxit, int64_t *m,
                    qbh_z *d_v, double *hessenberg, const char *purpose, qbh_solver_info *info);
/* The same recurrence on vectors stored as PACKED DOUBLES (d_v: two slots of n doubles, v[k-1], v[k] at (j%2)*n) for a
 * real operator on one GPU, purposes "sr_val0" and "dnmcs": nothing complex is allocated, so a sector whose complex
 * vectors would not fit (kagome 36 sites, Sz = 0: dim 9,075,135,300, 2 x 72.6 GB as doubles) runs on one MI355X with
 * a matrix-free operator.  qbh_vec_randomize_real fills a packed vector with the stream of vec_randomize. */
int qbh_lanczos_real_dev(const qbh_csr *A, int64_t k, int64_t np, int64_t maxit, int64_t *m,
                         double *d_v, double *hessenberg, const char *purpose, qbh_solver_info *info);
int qbh_vec_randomize_real(const qbh_csr *A, double *d_x, uint32_t seed);
/* eigenvec_CG (src/lanczos.cc:281-341) on four vectors stored as packed doubles; same contract as qbh_eigenvec_cg_dev. */
int qbh_eigenvec_cg_real_dev(const qbh_csr *A, int64_t maxit, int64_t *m, double E0, double *accu,
                             double *d_v, double *d_r, double *d_p, double *d_pp, qbh_solver_info *info);

/* Replaces eigenvec_CG<T,MAT> (src/lanczos.cc:281-341): CG on (H-E0)v = 0 with the
 * reference's restart/renormalise branch and the (machine_prec - E0) shift.  *m is in/out
 * (steps done), *accu out. */
int qbh_eigenvec_cg(const qbh_csr *A, int64_t maxit, int64_t *m, double E0, double *accu,
                    qbh_z *v_host, qbh_z *r_host, qbh_z *p_host, qbh_z *pp_host,
                    qbh_solver_info *info);
int qbh_eigenvec_cg_dev(const qbh_csr *A, int64_t maxit, int64_t *m, double E0, double *accu,
                        qbh_z *d_v, qbh_z *d_r, qbh_z *d_p, qbh_z *d_pp, qbh_solver_info *info);

/* Replaces iram<T,csr_mat<T>> / call_arpack (src/lanczos.cc:438-603) for order "sr"/"sa" (lowest) and
 * "lr"/"la" (highest) with the Krylov basis resident in HBM (thick-restart Lanczos == implicitly
 * restarted Lanczos for a Hermitian operator): nev wanted eigenpairs, ncv basis vectors
 * (nev + 2 <= ncv <= 64), at most maxit restarts, tol <= 0 meaning machine epsilon as in ARPACK
 * (src/lanczos.cc:452); the start vector is random (ARPACK info = 0, seed selects the Lehmer stream).
 * Outputs like iram: *nconv, eigenvals[nev] in the requested order, eigenvecs_host[nev*n] (may be
 * NULL).  info->n_reorth returns the number of restarts.  Other orders ("sm","lm") return
 * QBH_EUNSUPP: drive ARPACK over qbh_multmv for those. */
int qbh_iram(const qbh_csr *A, int64_t nev, int64_t ncv, int64_t maxit, const char *order, double tol,
             uint32_t seed, int64_t *nconv, double *eigenvals, qbh_z *eigenvecs_host, qbh_solver_info *info);

/* ---------------------------------------------- operator x vector (8f-3) --- */
/* Counterpart of model<T>::moprXvec_full (src/model.cc:1468-1538), vec_new = A |vec_old>, on device vectors in the
 * basis order of the device generators; the step that feeds lanczos(..., "dnmcs") in measure_full_dynamic
 * (src/model.cc:1696-1712).  Every target row gathers its contributions (no atomics, deterministic).
 * qbh_mopr_spin_dev: spin-1/2 sector with n_dn_old down spins (basis of qbh_gen_heisenberg).  kind 0: S^z_q = sum_s
 * coef[s] S^z_s (same sector); kind -1: S^-_q = sum_s coef[s] S^-_s (result lives in the sector with n_dn_old + 1 down
 * spins); kind +1: S^+_q (n_dn_old - 1).  d_vec_new must hold C(n_sites, n_dn_old - kind) elements.
 * qbh_mopr_onebody_dev: two-species fermions (basis of qbh_gen_hubbard), A = sum_k w[k] c+_{a[k], spin[k]} c_{b[k],
 * spin[k]} (spin 0 = up, 1 = down; a == b is the density n_{a,spin}); same sector. */
int qbh_mopr_spin_dev(int n_sites, int n_dn_old, int kind, const qbh_z *coef /* [n_sites] */, const qbh_z *d_vec_old,
                      qbh_z *d_vec_new, void *stream);
int qbh_mopr_onebody_dev(int n_sites, int n_up, int n_dn, int n_terms, const int32_t *a, const int32_t *b, const int32_t *spin,
                         const qbh_z *w, const qbh_z *d_vec_old, qbh_z *d_vec_new, void *stream);
/* The GENERAL form of moprXvec_full (src/model.cc:1468-1538): any mopr of the two families, written as a sum of ordered products
 * of elementary site operators,  A = sum_t coef[t] * O_{t,0} O_{t,1} ... O_{t,len_t-1}  (O_{t,0} the LEFTMOST factor; factors of
 * term t are entries term_ptr[t] .. term_ptr[t+1]-1 of op_kind / op_site / op_species; term_ptr[0] = 0).
 *   family 0, spin-1/2 sector with n_a_old down spins (basis of qbh_gen_heisenberg; n_b_old and op_species unused, may be 0 / NULL):
 *       op_kind 0 = S^z_s, 1 = S^+_s (down -> up), 2 = S^-_s (up -> down)
 *   family 1, two-species fermions with n_a_old up and n_b_old down particles (basis of qbh_gen_hubbard), species 0 = up, 1 = down:
 *       op_kind 0 = n_{s,sp}, 1 = c^dag_{s,sp}, 2 = c_{s,sp}; basis states are prod_{o ascending} c^dag_o |0> over the orbitals
 *       o = s + species * n_sites (all up operators left of all down operators), c^dag_o |w> = (-1)^{occupied orbitals below o} |w + o>
 * Any local 2 x 2 matrix of the reference's opr<T> is a combination of {1, S^z, S^+, S^-} resp. {1, n, c^dag, c}, so every
 * opr_prod / mopr over these site types is such a list.  Every product must change the particle numbers by the same amount (one
 * target sector: sec_new of the reference); *dim_new_out (may be NULL) = its dimension, d_vec_new must hold that many elements.
 * Every target row gathers its contributions (the adjoint factors run over the row's own pattern): no atomics, deterministic.
 * <= 4096 factors in all. */
int qbh_mopr_terms_dev(int family, int n_sites, int n_a_old, int n_b_old, int n_terms, const int32_t *term_ptr, const int32_t *op_kind,
                       const int32_t *op_site, const int32_t *op_species, const qbh_z *coef, const qbh_z *d_vec_old, qbh_z *d_vec_new,
                       int64_t *dim_new_out, void *stream);
/* Counterpart of model<T>::moprXvec_repr (src/model.cc:1715-1846) for S^z_q = sum_s coef[s] S^z_s between two momentum
 * sectors of the same n_dn (basis of qbh_gen_heisenberg_repr: all representatives, ascending; perms as there).  coef must
 * transform like a character, coef[g(s)] = eta(g) coef[s] (e.g. exp(i q.r_s)/sqrt(N)), and chars_new are the characters
 * of the TARGET momentum chi_new = chi_old * eta.  vec_new[a] = (sum_s coef[s] s^z_s(a)) * vec_old[a] for representatives
 * whose norm does not vanish at the target momentum, 0 for the others.  *dim_out (may be NULL) = number of representatives. */
int qbh_mopr_sz_repr_dev(int n_sites, int n_dn, int n_trans, const int32_t *perms, const double *chars_new,
                         const qbh_z *coef, const qbh_z *d_vec_old, qbh_z *d_vec_new, int64_t *dim_out);
/* The off-diagonal branch of moprXvec_repr (src/model.cc:1760-1830): S^-_q (kind -1, n_dn_old -> n_dn_old + 1) and S^+_q
 * (kind +1, n_dn_old -> n_dn_old - 1) from the momentum sector chars_old to chars_new = chars_old * eta, coef as above.
 * vec_new[b] = sum over (a, s): coef[s] chi_new(g*) sqrt(|S_b|/|S_a|) vec_old[a], g* the translation that takes the
 * flipped pattern to its representative b.  Contributions are accumulated with fp64 atomics (the reference uses a
 * critical section), so the result is reproducible only to rounding.  d_vec_new must hold the target sector's
 * representatives (*dim_new_out). */
int qbh_mopr_flip_repr_dev(int n_sites, int n_dn_old, int kind, int n_trans, const int32_t *perms, const double *chars_old,
                           const double *chars_new, const qbh_z *coef, const qbh_z *d_vec_old, qbh_z *d_vec_new,
                           int64_t *dim_old_out, int64_t *dim_new_out);

/* --------------------------------------------------------- checkpoints --- */
/* The reference's checkpoint files from the C ABI, host only.
 * qbh_vec_disk_write / _read: vec_disk_write / vec_disk_read (src/miscellaneous.cc:391-469): int64 n | n * elem_size
 * bytes | CRC-32 (boost::crc_32_type) of header + payload; read returns 0, or 1 where the reference returns 1.
 * qbh_crc32(running, data, nbytes): that CRC, start with 0.
 * qbh_ckpt_lanczos_update / _init: ckpt_lanczos_update / ckpt_lanczos_init for the "val" purposes
 * (src/ckpt.cc:23-297) on directory `dir` (the reference uses "out_Qckpt"), including the finish / rewind branches of
 * an interrupted update; *k_out = 0 means "nothing usable, start from scratch".  v / hessenberg: host arrays laid out
 * as lanczos() expects them.
 * qbh_lanczos_ckpt: lanczos(0, maxit - 1, ...) with enable_ckpt = true: resumes from `dir` when it holds a usable
 * step, commits a checkpoint every `every` steps (the two live Lanczos vectors are downloaded for it) and at the end;
 * max_steps > 0 stops after that many new steps; *converged (may be NULL) = the stop rule fired.
 * An update interrupted before its second marker is rewound to the last COMMITTED step (the reference's "one step back",
 * src/ckpt.cc:80-97, generalised to updates that are `every` steps apart); the two vectors are written under a temporary
 * name and renamed, so a committed file is never rewritten in place.  qbh_lanczos_ckpt needs the whole operator on one
 * GPU: a row shard returns QBH_EUNSUPP (every rank would write the same file names with its shard-local length). */
uint32_t qbh_crc32(uint32_t crc, const void *data, int64_t nbytes);
int qbh_vec_disk_write(const char *filename, int64_t n, int elem_size, const void *x);
int qbh_vec_disk_read(const char *filename, int64_t n, int elem_size, void *x);
int qbh_ckpt_lanczos_update(const char *dir, int64_t m, int64_t maxit, int64_t dim, int cnt_accuE0, double accuracy,
                            double theta0_prev, double theta1_prev, const qbh_z *v, const double *hessenberg,
                            const char *purpose);
int qbh_ckpt_lanczos_init(const char *dir, int64_t *k_out, int64_t maxit, int64_t dim, int *cnt_accuE0, double *accuracy,
                          double *theta0_prev, double *theta1_prev, qbh_z *v, double *hessenberg, const char *purpose);
int qbh_lanczos_ckpt(const qbh_csr *A, int64_t maxit, int64_t *m, qbh_z *v_host, double *hessenberg,
                     const char *purpose, int64_t every, int64_t max_steps, const char *dir, int *converged,
                     qbh_solver_info *info);
/* The CG half of the protocol (src/ckpt.cc:344-517; ckpt_CG_init / ckpt_CG_update / ckpt_CG_clean as eigenvec_CG calls them,
 * src/lanczos.cc:286,312,333): files CG_V<m>.dat, CG_R<m>.dat, CG_P<m>.dat in vec_disk_write format and the two markers
 * CG_updt.Qckpt1 / CG_updt.Qckpt2.  qbh_eigenvec_cg_ckpt is eigenvec_CG with enable_ckpt = true: resumes from `dir` when it
 * holds a step (otherwise starts from v with m = 0), commits a checkpoint every `every` steps and at the end, stops after
 * max_steps new steps when max_steps > 0; info->cg_resid[j] receives the residual of every step j made by this call (the rows
 * of log_CG.txt, src/lanczos.cc:308-311,334-337).
 * Row shards: qbh_lanczos_ckpt and qbh_eigenvec_cg_ckpt are COLLECTIVE under a communicator -- every rank checkpoints its slice
 * in dir/shard<r>of<P>/ with the same file names; the ranks meet between writing the new step and removing the old one, and
 * before a resume, so that all of them continue from the same step (<= 16 ranks). */
int qbh_ckpt_cg_update(const char *dir, int64_t m, int64_t dim, const qbh_z *v, const qbh_z *r, const qbh_z *p);
int qbh_ckpt_cg_init(const char *dir, int64_t *m_out, int64_t maxit, int64_t dim, qbh_z *v, qbh_z *r, qbh_z *p);
int qbh_ckpt_cg_clean(const char *dir);
int qbh_eigenvec_cg_ckpt(const qbh_csr *A, int64_t maxit, int64_t *m, double E0, double *accu, qbh_z *v_host, qbh_z *r_host,
                         qbh_z *p_host, qbh_z *pp_host, int64_t every, int64_t max_steps, const char *dir, int *converged,
                         qbh_solver_info *info);

/* Replaces hess_eigen (src/lanczos.cc:355-390), host only: eigen-decomposition of the
 * m x m tridiagonal held in hessenberg (ld = maxit), sorted by order ("sr","lr","sm","lm");
 * ritz[m], s[m*m] column-major. */
int qbh_hess_eigen(const double *hessenberg, int64_t maxit, int64_t m, const char *order,
                   double *ritz, double *s);

/* ------------------------------------------------------- communicator ---- */
/* Row sharding over P ranks, one process per GPU.  Rank r owns global rows
 * [r*nblk, min((r+1)*nblk, ncols)).  The host side (torch.distributed over RCCL in this
 * repo) owns the buffers and implements the two exchange steps the path has:
 *   allgather_x : d_xfull[q*nblk .. (q+1)*nblk) <- d_xsend of rank q, for all q
 *                 (a genuine shard, nrows < ncols, is split at creation into locally-owned and
 *                 remote columns; qbh_csr_download merges them back)
 *   allreduce   : in-place sum over ranks of d_scal[off .. off+n)
 * Both are enqueued on (or ordered with) the operator's stream and return 0 on success. */
typedef struct qbh_comm {
    int      rank, nranks;
    int64_t  nblk;            /* rows per rank block; d_xfull holds nranks*nblk elements    */
    qbh_z   *d_xsend;         /* [nblk]  device, owned by the host side                     */
    qbh_z   *d_xfull;         /* [nranks*nblk] device                                       */
    double  *d_scal;          /* [>= 16] device doubles                                     */
    double  *d_xfull_r;       /* [nranks*nblk] device doubles, or NULL: staging of the REAL wire format.
                                 When the operator and the vectors of a solve are real (exactly zero
                                 imaginary parts) only the real parts travel: the hook is called with
                                 packed = 1 and must gather nblk DOUBLES per rank from d_xsend (viewed as
                                 double[nblk]) into d_xfull_r; the library expands them into d_xfull.   */
    void    *ctx;
    int    (*allgather_x)(void *ctx, int packed);
    int    (*allreduce_sum)(void *ctx, int off, int n);
    /* optional split form of allgather_x (NULL = not provided): begin() enqueues the exchange and
     * returns, wait() orders the operator's stream after its completion.  When both are given, a row
     * shard applies its locally-owned columns between the two calls (overlap with xGMI traffic). */
    int    (*allgather_begin)(void *ctx, int packed);
    int    (*allgather_wait)(void *ctx);
    /* NULL: uniform blocks, rank q owns [q*nblk, (q+1)*nblk).  Otherwise the global row cuts of an nnz-balanced (ragged)
     * partition, row_cuts[0] = 0 <= ... <= row_cuts[nranks] = ncols (qbh_balanced_row_cuts): rank q owns
     * [row_cuts[q], row_cuts[q+1]), nblk is the longest block (the size of d_xsend), d_xfull / d_xfull_r hold ncols
     * elements indexed by GLOBAL column and the gather hook places rank q's block at offset row_cuts[q].  The array is
     * copied by qbh_csr_set_comm. */
    const int64_t *row_cuts;
    /* optional (NULL = not provided): the gather in PARTS.  A shard whose operator is split (qbh_opts.kron_split) sweeps the
     * gathered x band range by band range, and in the tiled block every rank sends a band range is a contiguous piece:
     * allgather_part_begin(ctx, part, nparts, off_len) enqueues, for every rank q, the exchange of the complex128 elements
     * [off_len[2q], off_len[2q] + off_len[2q+1]) of rank q's block (counted from the start of that block in d_xsend / d_xfull);
     * the parts are begun in ascending order right after one another.  allgather_part_wait(ctx, part) orders the operator's
     * stream after the completion of that part; the far pass of the band range runs while the later parts are on the wire. */
    int    (*allgather_part_begin)(void *ctx, int part, int nparts, const int64_t *off_len);
    int    (*allgather_part_wait)(void *ctx, int part);
    /* optional (NULL = not provided; ABI 600): the gather in parts with the wire format named -- packed = 1: the pieces are
     * DOUBLES [off_len[2q], + off_len[2q+1]) of rank q's block, from d_xsend viewed as double[] into d_xfull_r (qbh_opts.real_wire);
     * packed = 0: exactly allgather_part_begin.  Without it a real solve on split shards gathers in one piece (allgather_begin
     * with packed = 1). */
    int    (*allgather_part_begin_w)(void *ctx, int part, int nparts, const int64_t *off_len, int packed);
    /* optional (NULL = not provided; ABI 600): a PERSONALISED exchange in parts, for split shards that send every peer only the
     * major indices that peer reads (qbh_opts.sparse_gather).  Part `part` of `nparts`: to every peer q the elements
     * [send_off_len[2q], + send_off_len[2q+1]) of d_send, from every peer q into [recv_off_len[2q], + recv_off_len[2q+1]) of
     * d_recv (elements of elem_doubles doubles: 2 = complex128, 1 = real parts; library-owned device buffers; zero lengths are
     * skipped on both sides; the entries of the rank itself are 0).  Begun in ascending order like allgather_part_begin;
     * allgather_part_wait(ctx, part) orders the operator's stream after the completion of that part. */
    int    (*exchange_v)(void *ctx, int part, int nparts, const int64_t *send_off_len, const int64_t *recv_off_len, int elem_doubles,
                         const void *d_send, void *d_recv);
} qbh_comm;
int qbh_csr_set_comm(qbh_csr *A, const qbh_comm *comm);

/* The same two exchange steps implemented natively on RCCL (xGMI), no host-language callback in the SpMV loop: what a
 * C++ host such as the reference (one process per GPU, src/model.cc:1177-1181 calling lanczos()) links against.
 * qbh_rccl_unique_id: rank 0 obtains the 128-byte ncclUniqueId and hands it to the other ranks by whatever means the
 * host program has (MPI_Bcast, a file, torch.distributed).  qbh_comm_create_rccl: collective over all ranks; creates
 * the RCCL communicator, the exchange buffers, a side stream + events (the gather overlaps the locally-owned columns of
 * a split shard) and attaches them to the row-shard operator A.  row_cuts[nranks+1]: global row cuts (ragged allowed),
 * NULL = uniform blocks ceil(ncols / nranks).  qbh_comm_destroy detaches and frees (also done by qbh_csr_destroy). */
int qbh_rccl_unique_id(void *uid128);
int qbh_comm_create_rccl(qbh_csr *A, const void *uid128, int rank, int nranks, const int64_t *row_cuts);
int qbh_comm_destroy(qbh_csr *A);

/* ------------------------------------------------------------ stats ------ */
typedef struct qbh_stats {
    int64_t n_spmv;           /* SpMV launches since the last reset                       */
    double  ms_spmv;          /* sum of HIP-event kernel times (opts.profile != 0)        */
    double  ms_spmv_min;      /* fastest single launch                                    */
    int64_t n_gather;
    double  ms_gather;        /* time spent in allgather_x (event-timed)                  */
    int64_t n_spmv_real;      /* launches that used the real fast path (8-byte x gathers) */
} qbh_stats;
int qbh_get_stats(const qbh_csr *A, qbh_stats *s, int reset);
/* Block until everything enqueued on the operator's stream has finished (device building blocks
 * without a reduction are asynchronous; needed before another operator/stream touches the vectors). */
int qbh_sync(const qbh_csr *A);
/* Change one of the options that do not touch the stored form of the operator, after creation (everything else in qbh_opts is
 * fixed when the handle is made): "lanczos_pipeline", "profile", "tile_fold", "comm_reserve" (ABI 601; the same value on every
 * rank).  QBH_EINVAL for any other name. */
int qbh_csr_set_option(qbh_csr *A, const char *name, int value);
/* qbh_opts.major_partition: generator_major[i] = the major index the generator's ascending order gives the up configuration that
 * this operator holds at major index i (n_major entries, the same on every rank).  QBH_EUNSUPP for an operator in the generator's
 * own order. */
int qbh_csr_major_order(const qbh_csr *A, int32_t *generator_major, int64_t n_major);

/* ------------------------------------------------ synthetic operators ---- */
/* Measurement harness: device-side assembly of the benchmark Hamiltonians directly into
 * HBM (the reference's host pipeline, src/model.cc:619-685, cannot reach dim >= 1e8).
 * Fermi-Hubbard, H = -t sum_<ij>,s (c+_is c_js + h.c.) + U sum_i n_iup n_idn, on n_sites
 * sites with the given bond list (pairs, with multiplicity), N_up/N_dn fixed.
 * Basis index = rank(up config) * C(n_sites, n_dn) + rank(dn config) (colexicographic
 * ranks); fermion order: all up operators, then all down operators.  Builds FULL storage
 * for global rows [row_begin, row_end), columns ascending. */
int qbh_gen_hubbard(qbh_csr **out, int n_sites, int n_up, int n_dn, int n_bonds,
                    const int32_t *bonds /* [2*n_bonds] */, double t, double U,
                    int64_t row_begin, int64_t row_end, const qbh_opts *opts);
/* The same Fermi-Hubbard operator WITHOUT a stored matrix (counterpart of the matrix-free
 * model<T>::MultMv2, src/model.cc:941-1109): H = T_up (x) 1 + 1 (x) T_dn + U*D is applied from the two hop tables
 * (a few MB).  Same basis order and signs as qbh_gen_hubbard, so y = Hx is identical up to summation order; every
 * entry point that takes an operator handle works (SpMV, Lanczos, CG, IRAM, communicator); qbh_csr_download
 * returns QBH_EUNSUPP; qbh_csr_get_info reports the nnz a CSR of the operator would hold. */
int qbh_mf_hubbard(qbh_csr **out, int n_sites, int n_up, int n_dn, int n_bonds,
                   const int32_t *bonds /* [2*n_bonds] */, double t, double U,
                   int64_t row_begin, int64_t row_end, const qbh_opts *opts);
/* Spin-1/2 Heisenberg, H = J sum_<ij> S_i.S_j, fixed number of down spins; basis index =
 * colexicographic rank of the down-spin bit pattern. */
int qbh_gen_heisenberg(qbh_csr **out, int n_sites, int n_dn, int n_bonds,
                       const int32_t *bonds, double J,
                       int64_t row_begin, int64_t row_end, const qbh_opts *opts);
/* The same Heisenberg operator WITHOUT a stored matrix (matrix-free model<T>::MultMv2, src/model.cc:941-1109): every
 * row is unranked, its bonds flipped and the flipped patterns re-ranked on the fly from tables held in LDS.  Same
 * basis as qbh_gen_heisenberg; the dimension may exceed 2^31 (no column index is stored). */
int qbh_mf_heisenberg(qbh_csr **out, int n_sites, int n_dn, int n_bonds,
                      const int32_t *bonds, double J,
                      int64_t row_begin, int64_t row_end, const qbh_opts *opts);
/* Translation-symmetric sector of the same Heisenberg model, assembled on the device: counterpart of
 * model::generate_Ham_sparse_repr (src/model.cc:687-836).  The translation group is given explicitly: perms[g*n_sites
 * + s] = image of site s under translation g (g = 0 the identity, <= 64 translations), chars[2g], chars[2g+1] =
 * Re, Im of the momentum character chi_k(g) = exp(-i k.t_g).  Basis: ALL orbit representatives of the n_dn sector,
 * ascending; zero-norm representatives stay as decoupled rows with the fake diagonal fake_pos + i/dim (the
 * reference's convention, src/model.cc:735-740, default fake_pos 100).  Values are genuinely complex:
 * H[a][b] = sum h * conj(chi(g*)) * sqrt(|S_b|/|S_a|) (phase * sqrt(nu_i/nu_j) of src/model.cc:808-814).
 * The sector dimension is only known after the representatives have been enumerated, so a row shard is named by
 * (shard, n_shards): rows [shard*nblk, min(dim, (shard+1)*nblk)), nblk = ceil(dim / n_shards) -- the partition of
 * qbh_comm; (0, 1) = the whole sector.  *dim_out (may be NULL) receives the sector dimension.
 * When opts->value_dict is on and the sector holds at most 65536 distinct values, the value stream is emitted
 * directly as 1- or 2-byte codes: the 16 B/nnz complex128 array is never materialised, which is what lets the
 * 36-site Sz = 0 sector (nnz 1.4e10, 5712 distinct values) live on one GPU. */
int qbh_gen_heisenberg_repr(qbh_csr **out, int n_sites, int n_dn, int n_bonds, const int32_t *bonds, double J,
                            int n_trans, const int32_t *perms, const double *chars, double fake_pos,
                            int shard, int n_shards, int64_t *dim_out, const qbh_opts *opts);
/* The same with the caller's row cuts, row_cuts[n_shards + 1] rising from 0 to the sector dimension (ragged allowed; the
 * same array goes to qbh_comm.row_cuts / qbh_comm_create_rccl): shard q holds rows [row_cuts[q], row_cuts[q+1]).  Uniform
 * blocks of a momentum sector are unbalanced in work (4x5 half filling, 4 ranks: 63 to 98 ms per SpMV at +-5 % nonzeros):
 * time one SpMV per rank on uniform cuts, pass the times to quantum_basis_amd.dist.rebalance_cuts and generate again.
 * A first call with row_cuts = NULL (uniform) and dim_out returns the dimension the cuts have to add up to. */
int qbh_gen_heisenberg_repr_cuts(qbh_csr **out, int n_sites, int n_dn, int n_bonds, const int32_t *bonds, double J,
                                 int n_trans, const int32_t *perms, const double *chars, double fake_pos,
                                 int shard, int n_shards, const int64_t *row_cuts, int64_t *dim_out, const qbh_opts *opts);

/* The Hubbard family in a translation-symmetric (momentum) sector, assembled on the device: counterpart of
 * model::enumerate_basis_repr + generate_Ham_sparse_repr (src/model.cc:687-836) for two-species fermions, the path of
 * examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc.  The operator is
 *     sum_t ( amp_up[t] c^dag_{i_t,up} c_{j_t,up} + amp_dn[t] c^dag_{i_t,dn} c_{j_t,dn} )  +  U sum_i n_{i,up} n_{i,dn},
 * term_sites = (i_0, j_0, i_1, j_1, ...); terms on the same (i, j) are summed; i == j is a number operator.  It has to
 * commute with the translations: a Hamiltonian does; a one-body observable is translation-averaged first, as
 * model::measure_repr_static does (src/model.cc:1874-1888).  A non-Hermitian operator is accepted (rows are filled as
 * O[a][b]).  Density-density terms: pair_sites = (i_0, j_0, ...), pair_v[4p..4p+3] = the coefficients of n_{i,up} n_{j,up},
 * n_{i,up} n_{j,dn}, n_{i,dn} n_{j,up}, n_{i,dn} n_{j,dn} (extended Hubbard; with n_dn = 0 the spinless t-V model of
 * examples/trans_symmetric/latt_honeycomb/honeycomb_Spinless_Fermion.cc).  Spin exchange: exch_sites = (i_0, j_0, ...),
 * exch_amp[e] * (S+_i S-_j + S-_i S+_j).  no_double != 0 restricts the space to words without doubly occupied sites and
 * projects the hopping accordingly: with hops -t, exchange J/2 and pair_v = (0, -J/2, -J/2, 0) this is the t-J model of
 * examples/trans_symmetric/latt_kagome/kagome_tJ.cc.  perms / chars / fake_pos / shard / n_shards / dim_out as in
 * qbh_gen_heisenberg_repr.  Basis: all orbit
 * representatives of the words u | d << n_sites (operator order: all up, then all down), ascending; n_sites <= 31. */
int qbh_gen_hubbard_repr(qbh_csr **out, int n_sites, int n_up, int n_dn, int n_terms, const int32_t *term_sites,
                         const qbh_z *amp_up, const qbh_z *amp_dn, double U, int n_pairs, const int32_t *pair_sites,
                         const double *pair_v, int n_exch, const int32_t *exch_sites, const double *exch_amp, int no_double,
                         int n_trans, const int32_t *perms, const double *chars, double fake_pos,
                         int shard, int n_shards, int64_t *dim_out, const qbh_opts *opts);
/* ... with the caller's row cuts (see qbh_gen_heisenberg_repr_cuts) */
int qbh_gen_hubbard_repr_cuts(qbh_csr **out, int n_sites, int n_up, int n_dn, int n_terms, const int32_t *term_sites,
                              const qbh_z *amp_up, const qbh_z *amp_dn, double U, int n_pairs, const int32_t *pair_sites,
                              const double *pair_v, int n_exch, const int32_t *exch_sites, const double *exch_amp, int no_double,
                              int n_trans, const int32_t *perms, const double *chars, double fake_pos,
                              int shard, int n_shards, const int64_t *row_cuts, int64_t *dim_out, const qbh_opts *opts);

/* The same sector operator in MATRIX-FREE form with a small stored remainder (single GPU; no spin-exchange terms, real
 * up-species and number-operator amplitudes).  Representatives are ordered by the down pattern first: in a down block whose
 * pattern is trivially stabilised every up pattern is a representative, the up hops are the full-basis up-hop table applied
 * inside the block and each down hop is (target block, translation, coefficient); only the rows in or next to stabilised
 * blocks (< 1 %) are stored as CSR.  4x5 at half filling: ~40 MB of tables + ~3.4 GB instead of 364 GB.  The handle works with
 * every solver entry point; qbh_csr_download and qbh_csr_set_comm refuse it.
 * Row order ON THE DEVICE (qbh_opts.sector_orbit, default 1): inside a block the up patterns are held orbit by orbit of the
 * translation group -- a translated pattern then lies within n_trans rows of the same position in the target block, so a down
 * hop reads its target block once, front to back, and the up hops go through a per-ORBIT table.  Host vectors keep the order
 * of the representatives (every host seam translates); device vectors are in the handle's order
 * (qbh_csr_info.basis_internal = QBH_BASIS_SECTOR_ORBIT; qbh_vec_to_internal / qbh_vec_from_internal convert; vectors made by
 * qbh_vec_randomize(_real) hold the same stream as in the other order).  The construction is verified entry by entry at build;
 * translations that are not a group, up amplitudes that are not invariant under them, or more than 64 translations keep the
 * ascending order (basis_internal = 0). */
int qbh_mf_hubbard_repr(qbh_csr **out, int n_sites, int n_up, int n_dn, int n_terms, const int32_t *term_sites,
                        const qbh_z *amp_up, const qbh_z *amp_dn, double U, int n_pairs, const int32_t *pair_sites,
                        const double *pair_v, int n_trans, const int32_t *perms, const double *chars, double fake_pos,
                        int64_t *dim_out, const qbh_opts *opts);

/* moprXvec_repr (src/model.cc:1715-1846, diagonal branch) between two momentum sectors of qbh_gen_hubbard_repr for
 * O = sum_s ( coef_up[s] n_{s,up} + coef_dn[s] n_{s,dn} ), e.g. the density N_q (coef_up = coef_dn = e^{iq.r_s}) or S^z_q
 * (coef_up = -coef_dn = e^{iq.r_s}/2).  The coefficients must transform with a character, c_{g(s)} = eta(g) c_s (checked);
 * chars_new are the characters of the TARGET momentum chi_old * eta.  Both vectors are indexed like the rows of the sector
 * operators (all representatives, ascending) and live in HBM. */
int qbh_mopr_diag_hubrepr_dev(int n_sites, int n_up, int n_dn, int n_trans, const int32_t *perms, const double *chars_new,
                              const qbh_z *coef_up, const qbh_z *coef_dn, const qbh_z *d_vec_old, qbh_z *d_vec_new,
                              int64_t *dim_out);

/* moprXvec_repr (general branch) for the single-fermion operators  sum_s coef[s] c_{s,sigma}  (kind -1) and
 * sum_s coef[s] c^dag_{s,sigma}  (kind +1), species 0 = up / 1 = down, between the momentum sector (n_up_old, n_dn_old,
 * chars_old) and the sector with one particle less / more at chars_new = chars_old * eta, coef_{g(s)} = eta(g) coef_s: the
 * operators of the single-particle spectral function.  d_vec_new (dim_new elements, HBM) is overwritten. */
int qbh_mopr_c_hubrepr_dev(int n_sites, int n_up_old, int n_dn_old, int species, int kind, int n_trans, const int32_t *perms,
                           const double *chars_old, const double *chars_new, const qbh_z *coef, const qbh_z *d_vec_old,
                           qbh_z *d_vec_new, int64_t *dim_old_out, int64_t *dim_new_out);
/* Measurement harness (SURVEY 7 hard-part 1): the operator of qbh_gen_heisenberg (kind 0) / qbh_gen_hubbard (kind 1), built with
 * complex128 values on one GPU, re-expressed ON THE DEVICE in the reference's own basis order and fermion convention, i.e.
 * exactly the matrix the unchanged host code assembles (src/model.cc:619-685) at sizes that code cannot reach: basis sorted
 * by (odd sites, even sites) -- sort_basis_Lin_order, src/basis.cc:1144-1190, row index j = Lin_Ja[i_a] + Lin_Jb[i_b],
 * src/model.cc:665-670 -- electron states as two bits per site with operators ordered by site (src/basis.cc:2650-2664), so
 * H_ref = P D H_gen D P^T.  Full storage, columns ascending.  *out is a new handle; A is left untouched. */
int qbh_csr_reference_order(qbh_csr **out, const qbh_csr *A, int kind, int n_sites, int n_up, int n_dn, const qbh_opts *opts);
/* Copy the assembled shard back to host arrays (tests, CPU-baseline sample).  Any output
 * pointer may be NULL.  Rows [r0, r1) local to the shard; ia is rebased to 0.  An operator held in another order than the
 * caller's (qbh_csr_info.basis_internal != 0: a named / detected basis, a cut sector) returns the CALLER's rows, columns and
 * signs through its map (host-side from the whole internal operator: 20 B of host memory per nonzero, QBH_ENOMEM otherwise). */
int qbh_csr_download(const qbh_csr *A, int64_t r0, int64_t r1,
                     int64_t *ia /* [r1-r0+1] */, int32_t *ja, qbh_z *val);

#ifdef __cplusplus
}
#endif
#endif /* QBHIP_H */
