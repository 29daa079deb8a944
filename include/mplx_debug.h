/* mplx_debug.h -- diagnostics of libmplx.so: self-tests, statistics and a timing model.  Not part of the drop-in
 * boundary (include/mplx.h); exported by the same library for the tests and bench.py.  ABI v9 moved them here.      */
#ifndef MPLX_DEBUG_H
#define MPLX_DEBUG_H

#include "mplx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The host search's evaluation of a successor state (Primitive<Dim>(node, u, dt).evaluate(dt),
 * primitive.h:220-256, 321-331): node and out 4D+2 doubles, u one row of the control table.  Pure host
 * arithmetic (no device needed); the tests compare it with the oracle and with the device's states.   */
int mplx_selftest_forward_state(int32_t dim, int32_t control, const double *node, const double *u, double dt,
                                double *out);
/* Element-wise device evaluation of the libm-class operations the path uses,
 * for checking them against the host libm: op 0 a/b, 1 sqrt(a), 2 cos(a),
 * 3 sin(a), 4 round(a), 5 ceil(a).  Host pointers, n elements.               */
int mplx_selftest_math(mplx_ctx *ctx, int op, const double *a, const double *b, double *out,
                       int64_t n);
/* Statistics of the yaw pinning (mplx.h, "Yaw controls") since mplx_create: nodes flagged for the host-libm pass, fix
 * passes launched (both 0 on the BASELINE configurations).                                                        */
int mplx_yaw_pin_stats(const mplx_ctx *ctx, int64_t *flagged_nodes, int64_t *fix_passes);
/* ABI v6, diagnostic.  The list stores of an expansion launch on their own: for every node k the first count[k] entries
 * (rounded up to whole 128-byte lines as the kernels do) of every row present in d_lists are written with UNSPECIFIED
 * values, in the expansion kernels' order and with their store policy; count[] is read, not written.  A launch whose
 * lists stay in HBM is bound by exactly this once its arithmetic is cheaper (C4: DESIGN.md 5), and how long it takes
 * depends on the memory behind the allocation: bench.py times it on the lists of the timed launches
 * (roofline.store_only_ms).  Asynchronous on the context's stream.  OVERWRITES the successor entries.               */
int mplx_debug_store_model(mplx_ctx *ctx, const mplx_succ_lists *d_lists, int64_t n_nodes);

#ifdef __cplusplus
}
#endif
#endif
