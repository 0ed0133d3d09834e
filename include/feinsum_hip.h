/*
 * feinsum_hip.h -- C ABI of libfeinsum_hip.so, the MI355X (gfx950) evaluation
 * backend for the 3D DG-wave batched einsums (grad / div / face-mass).
 *
 * This is the drop-in boundary.  The reference (kaushikcfd/feinsum) has no FFI
 * seam of its own: its narrowest waist is the loopy executor call inside
 *   feinsum/measure.py:163-165  (validate_batched_einsum_transform)
 *   feinsum/measure.py:243-251,267 (timeit)
 * i.e. ``evt, outs = t_unit.executor(cq, **arg_dict)(cq, **arg_dict)`` --
 * "run the kernel generated for this BatchedEinsum on these device arrays,
 * asynchronously on this queue".  Every fe_*_f64 launcher below replaces that
 * call for one einsum family; fe_time_launches replaces the 5-launch timing
 * batches of measure.py:260-273; fe_device_info replaces ``cq.device.name`` +
 * feinsum/data/device_info.py:5-26.
 *
 * Conventions (all launchers):
 *   - all pointers are DEVICE pointers to C-contiguous float64 arrays;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream); no ownership moves: the caller keeps every
 *     buffer alive until it synchronises the stream; the launchers allocate
 *     nothing but the ticket-counter chunks of the dynamic walk (8.9 MB per
 *     sixteen streams, on demand, never inside a stream capture: see
 *     fe_stream_retired below);
 *   - inputs are read-only, outputs are fully overwritten (the reference's
 *     kernels assign, they do not accumulate: codegen/loopy.py:289-305);
 *   - E is the "long" element axis (feinsum SizeParam, einsum.py:26-41);
 *     E == 0 is a valid no-op;
 *   - return 0 on success, a negative FE_E* code otherwise; the message is
 *     available from fe_last_error() (thread-local).
 *   - thread-safety: re-entrant; distinct streams may be driven from distinct
 *     host threads, captured graphs replayed beside eager launches.  Mutable
 *     state: an init-once attribute cache, the record of prepared-operator
 *     buffers (fe_prepare_operator), and the ticket counters -- which belong to
 *     the launch's stream (or graph node), so that launches that can overlap
 *     never share any.
 *
 * `variant` selects the kernel implementation (the build's replacement for the
 * reference's transform archive lookup, sql_utils.py:247-294):
 *   FE_VARIANT_AUTO    best available kernel: MFMA where compiled, else tiled, else generic
 *   FE_VARIANT_GENERIC plain one-thread-per-output VALU kernel, any Np
 *   FE_VARIANT_MFMA    LDS-staged fp64-MFMA kernel (needs Np == 35 etc.;
 *                      FE_EUNSUPPORTED if the shape is not compiled)
 */
#ifndef FEINSUM_HIP_H
#define FEINSUM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FE_OK            0
#define FE_EINVAL       -1  /* bad shape / null pointer / negative size      */
#define FE_EUNSUPPORTED -2  /* (Np, variant, layout) not compiled            */
#define FE_EHIP         -3  /* HIP runtime error, text in fe_last_error()    */

#define FE_VARIANT_AUTO    0
#define FE_VARIANT_GENERIC 1
#define FE_VARIANT_MFMA    2
#define FE_VARIANT_TILED   3  /* LDS-tiled VALU kernel, any shape whose operator fits in LDS */
#define FE_VARIANT_MFMA_SPLIT 4 /* div of tetrahedra p = 1..4 only (fe_div3d_f64 and its _ex / _batched / _prepared forms):
                                 * the MFMA kernel walking both halves of the element range at once, so that the one
                                 * output array has two write windows.  Faster (4 % at E = 1e6) when the output lies
                                 * across a boundary between the two classes of physical memory (feinsum_amd/placement.py),
                                 * 1 % slower otherwise; results are those of FE_VARIANT_MFMA to the last bit or two.  Other entry points
                                 * answer FE_EUNSUPPORTED. */

/* families, for fe_time_launches / fe_flops_per_element */
#define FE_FAMILY_GRAD     1
#define FE_FAMILY_DIV      2
#define FE_FAMILY_GRADDIV  3
#define FE_FAMILY_FACEMASS 4
#define FE_FAMILY_DIVCOMP  5
#define FE_FAMILY_GRADPLANES 6
#define FE_FAMILY_MATAPPLY 7
#define FE_FAMILY_F32      0x100 /* or-ed into a family for fe_time_launches: the float32 launch (fe_launch_f32) */

/* face-mass operand layouts (bit flags) */
#define FE_FM_J_EF   0  /* J[E][nf]      (test_loopy_utils.py:41)            */
#define FE_FM_J_FE   1  /* J[nf][E]      (tuning/impls/ifj_fe_fej_to_ei.py)  */
#define FE_FM_R_FIJ  0  /* R[nf][Np][Nfp]                                    */
#define FE_FM_R_IFJ  2  /* L[Np][nf][Nfp]                                    */
#define FE_FM_R_T    4  /* last two operator axes swapped: R[nf][Nfp][Np], or
                           with FE_FM_R_IFJ  L[Nfp][nf][Np]  ('jfi',
                           tuning/impls/jfi_fe_fej_to_ei.py:46-56)           */

#define FE_MAX_FIELDS 8  /* fields per batched grad / div launch (more are split) */

/* operator flags for the _ex entry points of grad / div */
#define FE_OP_TRANSPOSED 1 /* D stored [3][Np(j)][Np(i)]: 'xre,rji,ej->xei',
                              'xre,rji,xej->ei' (tuning/impls/xre_rji_xej_to_ei_v1.py) */

/* ABI version: major*1000 + minor. */
int fe_version(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char* fe_last_error(void);

/* Number of visible HIP devices (<0: error code). */
int fe_device_count(void);

/* Device name + the peaks the roofline model uses (GFLOP/s fp64, GB/s HBM).
 * Replaces cq.device.name + data/device_info.py:5-26. */
int fe_device_info(int dev, char* name, size_t name_len,
                   double* peak_f64_gflops, double* peak_gbps);

/* grad:  out[x,e,i] = sum_{r,j} J[x,r,e] * D[r,i,j] * u[e,j]
 * 'xre,rij,ej->xei' (test/test_codegen.py:96-113); ndim = 3.
 *   J   [3][3][E]   D [3][Np][Np]   u [E][Np]   out [3][E][Np]              */
int fe_grad3d_f64(const double* J, const double* D, const double* u,
                  double* out, int64_t E, int32_t Np, int32_t variant,
                  void* stream);

/* grad / div with operator flags (FE_OP_*); flags == 0 is fe_grad3d_f64 / fe_div3d_f64. */
int fe_grad3d_f64_ex(const double* J, const double* D, const double* u,
                     double* out, int64_t E, int32_t Np, int32_t op_flags,
                     int32_t variant, void* stream);
int fe_div3d_f64_ex(const double* J, const double* D, const double* u,
                    double* out, int64_t E, int32_t Np, int32_t op_flags,
                    int32_t variant, void* stream);

/* div component:  out[e,i] = sum_{s,j} J[s,e] * D[s,i,j] * u[e,j]
 * 'se,sij,ej->ei' (test/test_codegen.py:34-66, one row of the batch;
 * tuning/impls/re_rij_ej_to_ei.py:147-157); op_flags: FE_OP_TRANSPOSED and
 * FE_OP_J_ES (J stored [E][3]: 'es,sij,ej->ei', examples/dg_wave_div.py:13-22).
 *   J [3][E]   D [3][Np][Np]   u [E][Np]   out [E][Np]                       */
#define FE_OP_J_ES 2
int fe_divcomp3d_f64(const double* J, const double* D, const double* u,
                     double* out, int64_t E, int32_t Np, int32_t op_flags,
                     int32_t variant, void* stream);

/* The same for ndim-dimensional simplices (ndim = 3 is fe_divcomp3d_f64; ndim = 2, triangles:
 * J [2][E] or [E][2], D [2][Np][Np]; MFMA kernels for Np in {3, 6, 10, 15, 21}, else the tiled kernel). */
int fe_divcomp_f64(const double* J, const double* D, const double* u,
                   double* out, int64_t E, int32_t ndim, int32_t Np, int32_t op_flags,
                   int32_t variant, void* stream);

/* div:   out[e,i] = sum_{x,r,j} J[x,r,e] * D[r,i,j] * u[x,e,j]
 * 'xre,rij,xej->ei' (tuning/impls/xre_rij_xej_to_ei.py:26-60).
 *   J [3][3][E]   D [3][Np][Np]   u [3][E][Np]   out [E][Np]                */
int fe_div3d_f64(const double* J, const double* D, const double* u,
                 double* out, int64_t E, int32_t Np, int32_t variant,
                 void* stream);

/* b fields through one grad / div launch, sharing J and D (the geometry factors are read once
 * per element instead of b times):
 *   'xre,rij,ej->xei' x b   (tuning/impls/batched_xre_rij_ej_to_xei.py)
 *   'xre,rij,xej->ei' x b   (tuning/impls/batched_xre_rij_xej_to_ei.py, _v2, _v3)
 *   u, out: HOST arrays of b >= 1 device pointers; shapes per field as in fe_grad3d_f64 /
 *   fe_div3d_f64.  b == 1 is fe_grad3d_f64_ex / fe_div3d_f64_ex; more than FE_MAX_FIELDS
 *   fields are split over several launches. */
int fe_grad3d_batched_f64(const double* J, const double* D,
                          const double* const* u, double* const* out,
                          int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                          int32_t variant, void* stream);
int fe_div3d_batched_f64(const double* J, const double* D,
                         const double* const* u, double* const* out,
                         int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                         int32_t variant, void* stream);

/* grad / div of ndim-dimensional simplices, b fields sharing J and D:
 *   J [ndim][ndim][E]   D [ndim][Np][Np]   grad: u_k [E][Np] -> out_k [ndim][E][Np]
 *                                           div:  u_k [ndim][E][Np] -> out_k [E][Np]
 * ndim == 3 is fe_grad3d_batched_f64 / fe_div3d_batched_f64; ndim == 2 (triangles: the same
 * transforms with ndim = 2, tuning/impls/xre_rij_ej_to_xei.py:20-22 `ndim = e.shape[0]`) has MFMA
 * kernels for Np in {3, 6, 10, 15, 21} and the LDS-tiled kernel for any other Np
 * (FE_VARIANT_GENERIC is not available for ndim == 2). */
int fe_grad_f64(const double* J, const double* D,
                const double* const* u, double* const* out,
                int64_t E, int32_t ndim, int32_t Np, int32_t b, int32_t op_flags,
                int32_t variant, void* stream);
int fe_div_f64(const double* J, const double* D,
               const double* const* u, double* const* out,
               int64_t E, int32_t ndim, int32_t Np, int32_t b, int32_t op_flags,
               int32_t variant, void* stream);

/* grad-type batch over separate geometry-factor arrays and output planes:
 *   out[3k + x][e,i] = sum_{r,j} J3[x][r,e] * D[r,i,j] * u[k][e,j],  k = 0..b-1, x = 0..2
 * i.e. the rows of a batched 're,rij,ej->ei' / 're,rji,ej->ei' that share u[k] and D are
 * evaluated together, D u[k] being formed once per field instead of once per row: the
 * curl-type batch of 12 rows over six fields and three J arrays in
 * tuning/impls/re_rji_ej_to_ei_3d_cross_product_v0.py:220-231 is one call with b = 6.
 *   J3:  HOST array of 3 device pointers, each [3][E]
 *   u:   HOST array of b device pointers, each [E][Np]
 *   out: HOST array of 3 b device pointers, each [E][Np] or NULL (plane not wanted); every
 *        field needs the same number (1..3) of non-NULL planes. */
int fe_gradplanes3d_f64(const double* const* J3, const double* D,
                        const double* const* u, double* const* out,
                        int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                        int32_t variant, void* stream);

/* element-local operator, b fields sharing J and D:
 *   out_k[e,i] = J[e] * sum_j D[i,j] * u_k[e,j]      'e,ij,ej->ei' x b
 *                                                    (tuning/impls/e_ij_ej_to_ei_no_prftch.py:30-38)
 *   J == NULL:   out_k[e,i] = sum_j D[i,j] u_k[e,j]  'ij,ej->ei' (tuning/impls/ij_ej_to_ei_no_prftch.py)
 *   J [E] or NULL   D [Np][Np] ([Np(j)][Np(i)] with FE_OP_TRANSPOSED)   u_k, out_k [E][Np]
 *   u, out: HOST arrays of b device pointers.  MFMA kernels for Np in {3,4,6,10,15,20,35,56}. */
int fe_matapply_f64(const double* J, const double* D,
                    const double* const* u, double* const* out,
                    int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                    int32_t variant, void* stream);

/* fused grad + div sharing J and D (BASELINE config 3), one persistent launch:
 *   grad_out[3][E][Np] from u_grad[E][Np];  div_out[E][Np] from v_div[3][E][Np] */
int fe_graddiv3d_f64(const double* J, const double* D,
                     const double* u_grad, const double* v_div,
                     double* grad_out, double* div_out,
                     int64_t E, int32_t Np, int32_t variant, void* stream);

/* The three einsums of one DG wave operator evaluation -- div(v), grad(u) and the lift of b
 * face fields -- in ONE persistent launch (examples/wave_3d_p4_auto.py:16-63 runs them as
 * three kernels separated by global barriers; they share no data except J and D).
 * Arguments as in fe_graddiv3d_f64 and fe_facemass_f64 (Jface, R, f, lift, fm_layout_flags).
 * Single launch for the tetrahedral orders p = 1..4 with 2 <= b <= 4; anything else runs
 * as fe_graddiv3d_f64 followed by fe_facemass_f64. */
int fe_waveop3d_f64(const double* J, const double* D,
                    const double* u_grad, double* grad_out,
                    const double* v_div, double* div_out,
                    const double* Jface, const double* R,
                    const double* const* f, double* const* lift,
                    int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b,
                    int32_t fm_layout_flags, int32_t variant, void* stream);

/* face-mass (lift), b fields sharing J and R:
 *   out_k[e,i] = sum_{f,j} J[e,f] * R[f,i,j] * v_k[f,e,j],  k = 0..b-1
 * 'ef,fij,fej->ei' x b (test/test_loopy_utils.py:34-48) or, with layout flags,
 * 'ifj,fe,fej->ei' (tuning/impls/ifj_fe_fej_to_ei.py:46-60).
 *   v, out: HOST arrays of b device pointers; v_k [nf][E][Nfp]; out_k [E][Np] */
int fe_facemass_f64(const double* J, const double* R,
                    const double* const* v, double* const* out,
                    int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b,
                    int32_t layout_flags, int32_t variant, void* stream);

/* ---- prepared operators ------------------------------------------------------------------
 * The operator matrices (D of grad / div, R or L of face-mass) are constant across the launches
 * of a time-stepping code, while every launch rebuilds their MFMA register fragments from the
 * plain array in its prologue.  fe_prepare_operator writes an operator ONCE in the kernels'
 * fragment layout; the *_prepared_* launchers then fetch the fragments with coalesced loads
 * instead (no LDS staging, no block barrier).  It plays the role of the operator prefetch of the
 * reference's transforms (tuning/impls/xre_rij_ej_to_xei.py:26-275, `prftch_u_to_local` and the
 * D slab in __local) -- hoisted out of the launch altogether.
 * MEASURED: launch times are the same with and without it (grad 0.2131 vs
 * 0.2142 ms at E = 1e6, 25.0 vs 26.7 us at the reference's default E = 1e5): the prologue already
 * overlaps the first tiles' load latency, which is what a launch waits for.  Nothing in the package
 * uses prepared operators by default; the entry points are kept for callers whose launch sequence
 * is prologue bound (many tiny batches).
 *
 *   family    FE_FAMILY_GRAD, FE_FAMILY_DIV or FE_FAMILY_GRADDIV: `op` = D[3][Np][Np]
 *             (FE_OP_TRANSPOSED in `flags` for [3][Np(j)][Np(i)]); one buffer serves grad AND div.
 *             FE_FAMILY_FACEMASS: `op` = R / L with FE_FM_R_IFJ / FE_FM_R_T in `flags`.
 *   prepared  caller-owned DEVICE buffer of FE_PREPARED_OPERATOR_BYTES bytes, 16-byte aligned;
 *             written asynchronously on `stream`.
 * Tetrahedra p = 1..4 (FE_EUNSUPPORTED otherwise: launch without a prepared operator).
 *
 * INVALIDATION IS THE CALLER'S: the buffer is a snapshot of `op`.  After changing the operator's
 * values call fe_prepare_operator again (same buffer is fine) before the next prepared launch;
 * the launchers cannot see that the snapshot is stale.  They do refuse (FE_EINVAL) a buffer that
 * this process did not prepare, or prepared for another shape / layout flag -- a host-side record
 * keyed by the buffer address, the library's only mutable state besides the init-once kernel
 * attribute cache (mutex protected; prepare a buffer before handing it to other threads).
 * A prepared launcher given NULL, or running a kernel without a prepared form (tiled / generic
 * variants, multi-plane grad batches), behaves exactly like its plain counterpart; results are
 * bitwise identical either way. */
#define FE_PREPARED_OPERATOR_BYTES (96 * 1024)
int fe_prepare_operator(int32_t family, const double* op, int32_t Np, int32_t nf, int32_t Nfp,
                        int32_t flags, void* prepared, void* stream);
/* The record of a prepared buffer also holds the address of the operator array it is a snapshot of and the device:
 * a *_prepared_* launch whose operator argument is another array (or that runs on another device) is refused with
 * FE_EINVAL instead of silently computing with the snapshot.  Call fe_release_prepared BEFORE freeing a prepared buffer:
 * it drops the record, so that an unrelated later allocation at the same address is not taken for a prepared operator
 * (FE_EINVAL if `prepared` has no record). */
int fe_release_prepared(const void* prepared);

/* fe_grad3d_batched_f64 / fe_div3d_batched_f64 / fe_facemass_f64 / fe_graddiv3d_f64 /
 * fe_waveop3d_f64 with the operator(s) ALSO given in prepared form (or NULL). */
int fe_grad3d_prepared_f64(const double* J, const double* D, const void* D_prepared,
                           const double* const* u, double* const* out,
                           int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                           int32_t variant, void* stream);
int fe_div3d_prepared_f64(const double* J, const double* D, const void* D_prepared,
                          const double* const* u, double* const* out,
                          int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                          int32_t variant, void* stream);
int fe_facemass_prepared_f64(const double* J, const double* R, const void* R_prepared,
                             const double* const* v, double* const* out,
                             int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b,
                             int32_t layout_flags, int32_t variant, void* stream);
int fe_graddiv3d_prepared_f64(const double* J, const double* D, const void* D_prepared,
                              const double* u_grad, const double* v_div,
                              double* grad_out, double* div_out,
                              int64_t E, int32_t Np, int32_t variant, void* stream);
int fe_waveop3d_prepared_f64(const double* J, const double* D, const void* D_prepared,
                             const double* u_grad, double* grad_out,
                             const double* v_div, double* div_out,
                             const double* Jface, const double* R, const void* R_prepared,
                             const double* const* f, double* const* lift,
                             int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b,
                             int32_t fm_layout_flags, int32_t variant, void* stream);

/* Text table of the MFMA / tiled kernels configured so far in this process, one line each:
 * threads, VGPRs, scratch, LDS per block and the resident blocks per CU the HIP occupancy query
 * reports next to the number the launch geometry assumes.  (A kernel is configured -- and the
 * residency checked: FE_EHIP if it fell below what the persistent grid is sized for -- on its first
 * launch on a device.)  Returns the full length of the table; at most buf_len - 1 characters are
 * copied.  No reference counterpart: the reference reads such figures off loopy's generated code. */
int fe_kernel_resources(char* buf, size_t buf_len);

/* ---- split allocator: arrays backed by ALTERNATING pieces of two classes of physical memory -------------------
 * New functionality; it replaces the per-array allocation of the reference's timing path (cla.zeros / cl.array.to_device
 * through a PyOpenCL MemoryPool, src/feinsum/measure.py:44-60,80-108,236-246) for arrays a launch WRITES.
 * On MI355X write streams confined to one class of physical memory are ~25 % slower than streams split over two
 * (DESIGN.md section 3d); the DG launches gain 5-14 % when every stream's write window is spread over two classes.
 * fe_split_alloc returns a virtually contiguous device array of `bytes` bytes on the current device whose 4 MiB pieces
 * (one physical handle each) alternate between two classes; the class of memory is measured when the pool obtains it
 * (groups of 32 pieces, a two-stream write probe against a reference group of every class seen so far).
 * No timing scan, no arena: mapped memory = bytes rounded up to 2 MiB.  Arrays below 8 MiB are one plain handle.
 * Address ranges are never re-used within a process (ROCm 7.2 keeps translating a re-mapped range to its first handle:
 * tools/vmm_remap_test.cpp); a freed array gives its memory back, not its (plentiful) address space.
 * The pointer is an ordinary device pointer for kernels, copies and the launchers of this library; it must be released
 * with fe_split_free (not hipFree), which waits for the device like hipFree does and returns the pieces to the pool.
 * Synchronous host calls, serialised per device; `flags` is reserved (0).
 *   fe_split_info   JSON about one array: {"bytes", "mapped_bytes", "piece_mib", "pieces", "pieces_by_class" (class ids
 *                   in order of discovery), "first_pieces": the classes of the first 16 pieces, "tail_bytes": the last,
 *                   unclassified handle, "alloc_ms"}
 *   fe_split_stats  JSON about the current device's pool: classes seen, free pieces per class, pieces created, groups
 *                   probed, spacer bytes used to skip runs of one class, milliseconds spent; "unsplit_refused" counts
 *                   the arrays fe_split_alloc REFUSED (FE_EUNSUPPORTED) because no second class was found within the
 *                   budget: all pieces from one class is the worst placement there is, worse on average than an ordinary
 *                   allocation -- allocate such an array ordinarily (round 5; FEINSUM_SPLIT_UNSPLIT=1 restores round 4's
 *                   behaviour, counted in "unsplit_arrays")
 *   fe_split_reserve announces the total size of the arrays about to be allocated: half of it is collected of each of two
 *                   classes at once, while the search through the driver's memory is still inside the first class (arrays
 *                   allocated one by one otherwise take only what each needs -- at least 512 MiB per class are collected
 *                   anyway -- and a class the search has left does not come back); allocates nothing that stays
 *   fe_split_trim   releases the pool's free pieces to the driver
 * Both JSON calls return the length written (>= 0) or a negative error code. */
int fe_split_alloc(void** ptr, size_t bytes, int32_t flags);
int fe_split_free(void* ptr);
int fe_split_info(const void* ptr, char* buf, size_t buf_len);
int fe_split_stats(char* buf, size_t buf_len);
int fe_split_reserve(size_t bytes);
int fe_split_trim(void);

/* Algorithmic flops per element for a family (numerator of GFLOP/s; same
 * counter as measure.py:278-331 on the opt_einsum-optimal schedule):
 * grad/div 2*3*Np*Np + 2*9*Np; face-mass b*(nf*Nfp + 2*Np*nf*Nfp);
 * div component 3*Np + 2*3*Np*Np (grad planes: the same per output plane, b = planes);
 * element-local operator b*(2*Np*Np + Np). */
int64_t fe_flops_per_element(int32_t family, int32_t Np, int32_t nf,
                             int32_t Nfp, int32_t b);

/* Argument pack for fe_time_launches (one struct for every family; unused
 * fields are ignored). */
typedef struct fe_argpack {
    const double* J;
    const double* D;          /* D (grad/div) or R/L (face-mass)              */
    const double* u;          /* grad: u[E][Np]; div: u[3][E][Np]             */
    const double* v_div;      /* graddiv only                                 */
    double* out;              /* grad / div output; graddiv: grad_out         */
    double* out2;             /* graddiv: div_out                             */
    const double* const* v;   /* face-mass, batched grad / div (b > 1): inputs  */
    double* const* outs;      /* ... and outputs (host arrays of b ptrs)        */
    int64_t E;
    int32_t Np, nf, Nfp, b, layout_flags, variant;   /* layout_flags: FE_FM_* or FE_OP_* by family */
    const double* const* j3;  /* FE_FAMILY_GRADPLANES: 3 ptrs; v = u (b), outs = planes (3 b) */
    int32_t ndim;             /* grad / div: 0 or 3 = tetrahedra, 2 = triangles (v / outs, any b) */
    const void* prepared;     /* the operator D (or R) in prepared form, or NULL (fe_prepare_operator) */
} fe_argpack;

/* float32 operands (the reference validates float32 einsums at 1e-6 and carries float32 peaks:
 * src/feinsum/measure.py:178-192, src/feinsum/data/device_info.py:5-26).  One entry point for the families
 * FE_FAMILY_GRAD / DIV / DIVCOMP / MATAPPLY / FACEMASS: every pointer of `args` is read as float (the struct's
 * `double` types are nominal here), layouts, `ndim` and `layout_flags` as for the float64 entry points; `prepared` is
 * ignored.  grad of tetrahedra p = 1 ... 4 (Np = 4, 10, 20, 35) and div and face-mass at p = 4 (Np = 35, ndim 3; face-mass:
 * nf = 4, Nfp = 15, any b, every layout) run on v_mfma_f32_16x16x4_f32 (fe_grad_f32.h, fe_div_f32.h, fe_facemass_f32.h) when
 * their operands are 16-byte aligned and E is a multiple of 4 (`variant` FE_VARIANT_TILED forces the tiled kernel); grad walks
 * dynamically and hints its loads like the float64 kernels (fe_set_tail_rounds, fe_set_temporal_loads_mib below); everything
 * else runs on the LDS-tiled VALU kernel in float (fe_tiled.h), any shape whose operator fits in LDS (FE_EUNSUPPORTED
 * otherwise). */
int fe_launch_f32(int32_t family, const fe_argpack* args, void* stream);

/* Dynamic walk of the persistent kernels (feinsum_amd/csrc/fe_common.h): behind two statically walked rounds the tiles are
 * handed to the waves by tickets.  Sets the largest number of full rounds handed out that way for later launches of this
 * process (default: all, or the environment's FEINSUM_TAIL_ROUNDS; negative = static walk) and returns the previous value.
 * A tuning knob: results do not depend on it (every tile's arithmetic is position independent). */
int fe_set_tail_rounds(int32_t rounds);
/* Launches of fewer than `rounds` full rounds of tiles walk statically (default 4, with the four-and-a-half-rounds rule of
 * DESIGN.md section 3f; also FEINSUM_TAIL_MIN_ROUNDS; at least 2).  Returns the previous setting.  A tuning knob. */
int fe_set_tail_min_rounds(int32_t rounds);

/* Who owns ticket counters (the re-entrancy contract of SURVEY 8(b): no launch shares mutable state with a launch that can
 * run beside it).  A launch's counters are zero before and after it, so only launches the device serialises share them:
 * an eager launch uses the counter group of its STREAM (the per-thread default stream: of its thread), a launch recorded
 * during stream capture gets a group of its own until fe_graph_retired (the graph may be replayed beside anything).
 * Any number of streams, driven from any number of host threads, may launch at the same time.  Groups (0.56 MB) are
 * allocated on demand outside capture, at most 256 per device; a launch that finds none walks statically (same results).
 *   fe_stream_retired  a stream was destroyed (its launches have completed): its group may serve another stream.
 *                      Returns 1 if the stream had one, 0 if not.  Optional -- without it the group stays with the handle
 *                      VALUE: call it before hipStreamDestroy of a stream with work still in flight, or a new stream that is
 *                      given the same handle could run beside that work on the same counters.
 *   fe_tail_check      waits for the current device, then counts the non-zero words in all counter groups (there must be
 *                      none: a launch that did not run to completion would leave some, and later launches through that
 *                      group would skip tiles) and, with `repair`, zeroes them.  Also reports the groups allocated, the
 *                      streams that own one and the groups given to captured launches (any out pointer may be NULL).
 *                      FEINSUM_TAIL_CHECK=1 runs the same check on the launch's group before every dynamic launch
 *                      (synchronises the stream: a debugging aid).
 * The reference's executor is single-queue (src/feinsum/measure.py:163-165,243-251); no counterpart. */
int fe_stream_retired(void* stream);
int fe_tail_check(int32_t repair, int64_t* dirty_words, int32_t* groups, int32_t* streams, int32_t* captured);
/* Launches recorded during stream capture (round 5):
 *   ONE EXECUTABLE PER CAPTURE.  A captured launch's counter group is part of the graph NODE, so every executable
 *   instantiated from one captured graph uses the same counters, and HIP orders only the launches of ONE executable against
 *   each other: two executables of one capture must never run at the same time (capture again for a second executable).
 *   fe_capture_id     *id = the id of the capture in progress on `stream` (hipStreamGetCaptureInfo), 0 when it is not capturing:
 *                     ask between the begin and the end of the capture, keep it with the graph.
 *   fe_graph_retired  the graph captured under `capture_id` and all its executables have been destroyed (their launches have
 *                     completed): the groups its nodes owned may serve others.  Returns how many came back (0: unknown id).
 *                     Without it a process that captures again and again runs out of groups after 256 captured launches and
 *                     walks statically from then on (3-5 % slower, never wrong).
 *   fe_tail_stats     out[0..n): groups allocated, streams owning one, groups owned by graph nodes, spare groups, `exhausted`
 *                     (launches that found every group owned), `static_fallbacks` (launches that wanted tickets and walked
 *                     statically for any reason), failed chunk allocations (retried later), groups verified / repaired after
 *                     an FE_EHIP return, live captures, the cap (256), FE_EHIP returns so far.  Returns the number written.
 * After ANY FE_EHIP return of the process the first dynamic launch of every stream waits for that stream once and verifies
 * (repairs) its group: a launch that did not run to completion cannot leave tickets behind for later launches. */
#define FE_TAIL_STATS 12
int fe_capture_id(void* stream, uint64_t* id);
int fe_graph_retired(uint64_t capture_id);
int fe_tail_stats(int64_t* out, int32_t n);
/* Test hook: waits for the device and writes `value` into one ticket counter of the group `stream` owns (FE_EINVAL if it
 * owns none) -- the state an interrupted launch would leave behind; tests/test_gpu_streams.py shows fe_tail_check finding
 * and repairing it. */
int fe_tail_plant(void* stream, uint32_t value);

/* Loads of the streamed operand (u / v) are non-temporal -- every byte is read once per launch -- unless the launch's inputs
 * are at most `mib` MiB (default 248, also FEINSUM_TEMPORAL_LOADS_MIB; 0 = never): then they are plain loads that may stay in
 * the 256 MiB Infinity Cache, where the next launch on the same arrays finds them (grad at the reference's default E = 1e5,
 * src/feinsum/measure.py:202: 24.8 -> 23.1 us; above the cache size plain loads cost 7-12 %; div and face-mass launches
 * switch only above 80 / 64 MiB of inputs, below which plain loads measured slower; the families whose footprint is mostly inputs
 * keep plain loads beyond `mib`, where they still gain -- div up to 280/248 of it, div + grad 310/248, launches with face-mass in
 * them 320/248: the pipeline at E = 1e5, inputs 307 MiB, 99.6 -> 94.8 us; 1048576 or more = always, for A/B runs).  Stores are non-temporal at every
 * size.  Applies to the MFMA kernels of the orders p = 1..4 (float64).  Returns the previous setting.  Results do not depend
 * on it. */
int fe_set_temporal_loads_mib(int32_t mib);

/* Output stores are non-temporal -- except in grad launches (p = 1..4, one field, static walk) that write at most `mib` MiB
 * (default 176, also FEINSUM_WRITE_THROUGH_MIB; 0 = never): those store write-through (sc0 sc1), which leaves no dirty lines for
 * the end of the launch to write back (E = 1e5: 24.0 -> 23.3 us).  Returns the previous setting.  Results do not depend on it. */
int fe_set_write_through_mib(int32_t mib);

/* What the launcher decided for the MFMA launch this THREAD enqueued last (p = 1..4 grad / div / face-mass and the fused
 * launches; other paths leave it unchanged): out[0..n) = {valid, dynamic walk (tickets behind the static rounds), plain
 * (temporal) loads of the streamed operand, write-through stores, blocks, waves per block, kernel kind (bit 2: div with the
 * interleaved B build; bit 3: div or grad with a quarter-tile tail; bit 4: grad with a staggered start), bodies of a fused launch, tiles (summed over the bodies), statically walked tiles}.  Returns the number
 * written.  For reports (bench.py prints these instead of re-deriving the launcher's rules). */
#define FE_LAST_LAUNCH_INFO 10
int fe_last_launch_info(int64_t* out, int32_t n);

/* div launches (p = 4, one field; static or dynamic walk) of at most `tiles` tiles run on the kernel that builds its B fragments
 * k-quad by k-quad between the MFMA groups of the same wave (default 37500 = E 6e5: -3 ... -6 % at E = 8e4 ... 3e5, level above;
 * 0 = never; also FEINSUM_DIV_INTERLEAVE_TILES).  Returns the previous setting.  Bitwise the results of the plain kernel. */
int64_t fe_set_div_interleave(int64_t tiles);
/* ... and, under the static walk with one field, the tiles behind the last full round -- when they fill at most an eighth of a
 * round -- as quarter tiles of four elements, one per wave (default on: E = 1e5 -4 %; also FEINSUM_DIV_QUARTER_TAIL).  Returns
 * the previous setting.  Bitwise the results of the plain kernel. */
int fe_set_div_quarter_tail(int32_t on);
/* The same for grad launches of one field on the static walk (tetrahedra p = 4): the four elements of a quarter tile run stage 1
 * on v_mfma_f64_4x4x4_4b with the fragments of the full tiles and stage 2 through LDS (default on; also
 * FEINSUM_GRAD_QUARTER_TAIL; a value n >= 4 widens the rule from an eighth to 1/n of a round, for measurements: a quarter of a
 * round loses 1 - 5 %).  Returns the previous setting.  Bitwise the results of the full tiles. */
int fe_set_grad_quarter_tail(int32_t on);
/* Short grad launches of one field on the static walk (tetrahedra p = 4, 2.5 to 4.5 rounds of tiles): the blocks on every second
 * CU of an XCD start about 2.4 us -- half a SIMD's tile period -- late, so that the chip's two halves do not issue their stores
 * in step (default on: E = 8.2e4 ... 1.47e5 -0.5 ... -4 %; also FEINSUM_GRAD_STAGGERED_START).  Returns the previous setting.
 * Results do not depend on it. */
int fe_set_grad_staggered_start(int32_t on);
/* Phase priorities in the eight-wave kernels of tetrahedra p = 5 (grad, div): the waves' f64 VALU phases at raised issue
 * priority, their matrix phases at priority 0 (default off: -1 % for div at E >= 1e6, +-1 % for grad; also
 * FEINSUM_PHASE_PRIORITY_P5).  Returns the previous setting.  Results do not depend on it. */
int fe_set_phase_priority_p5(int32_t on);

/* Size the persistent grids as if the device had `cus` compute units (0 = what the device reports; also
 * FEINSUM_CU_LIMIT).  MI355X partitions report 32 (CPX) or 64 (QPX) CUs: grids of fewer than 128 blocks walk statically
 * (a ticket pool is drained by the blocks b with (b / 8) % 16 == pool).  Returns the previous limit.  Results do not
 * depend on it. */
int fe_set_cu_limit(int32_t cus);

/* Enqueue n_launches back-to-back launches of `family` on `stream`, bracketed
 * by HIP events recorded on that same stream; blocks until the last one is
 * done and returns the elapsed milliseconds of the whole batch in *ms_out.
 * Mirrors the 5-launch batches of measure.py:260-273 (evt.wait() fences). */
int fe_time_launches(int32_t family, const fe_argpack* args, int32_t n_launches,
                     void* stream, float* ms_out);

/* ---- generic einsum (any explicit-mode subscripts; float64 or float32) ----
 * The device restatement of the loop nest generate_loopy emits for the trivial
 * schedule (codegen/loopy.py:242-305): one output entry per thread,
 *   out[o...] = sum_{s...} prod_p operand_p[o..., s...].
 * Used for einsums outside the DG families (e.g. test/test_measure.py:33-52).
 * Strides are in ELEMENTS; 0 for an index an operand does not carry.  The output
 * is C-contiguous in out_extent order. */
#define FE_MAX_EINSUM_OPERANDS 8
#define FE_MAX_EINSUM_INDICES  8
#define FE_DTYPE_F64 0
#define FE_DTYPE_F32 1

typedef struct fe_einsum_desc {
    int32_t n_operands, n_out, n_sum, dtype;
    int64_t out_extent[FE_MAX_EINSUM_INDICES];
    int64_t sum_extent[FE_MAX_EINSUM_INDICES];
    int64_t op_out_stride[FE_MAX_EINSUM_OPERANDS][FE_MAX_EINSUM_INDICES];
    int64_t op_sum_stride[FE_MAX_EINSUM_OPERANDS][FE_MAX_EINSUM_INDICES];
} fe_einsum_desc;

typedef struct fe_einsum_ptrs {
    const void* p[FE_MAX_EINSUM_OPERANDS];
} fe_einsum_ptrs;

/* operands: HOST array of n_operands device pointers. */
int fe_einsum_generic(const fe_einsum_desc* desc, const void* const* operands,
                      void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FEINSUM_HIP_H */
