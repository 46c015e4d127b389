// PRODUCT and DIAGNOSTICS builds of libisr_sr.
//
//   make            -> ../lib/libisr_sr.so        the library a deployment ships: no `isrDebug*` export, no ablation switch, no stamp buffer, no fault
//                                                 injection; the parameter blocks' diagnostic fields are compile-time constants (0 / NULL / -1), so every
//                                                 `p.dbg & ...`, `if (p.stamps)`, `p.faultTile == ...` in a kernel folds away -- the hot loops test nothing;
//                                                 the experimental kernel forms (sr_conv_ups4.h, _ups5.h, _upsp.h, _ups4r.h, _upsw.h) are not compiled in.
//   make diag       -> ../lib/libisr_sr_diag.so   the same sources with -DISR_DIAG=1: the fields are real members, the `isrDebugSet*` switches exist, the
//                                                 experimental forms are selectable (tools/, the timeout-path and form-parity tests: ops.diagnostics_library()).
//
// A diagnostic member is declared with ISR_DIAG_MEMBER(type, name, off) and written by the host with ISR_DIAG_SET(lvalue, value).
#pragma once
#ifdef ISR_DIAG
#define ISR_DIAG_MEMBER(type, name, off) type name
#define ISR_DIAG_SET(lvalue, value) (lvalue) = (value)
#define ISR_DIAG_ON 1
#else
#define ISR_DIAG_MEMBER(type, name, off) static constexpr type name = off
#define ISR_DIAG_SET(lvalue, value) ((void)0)
#define ISR_DIAG_ON 0
#endif

#include <cstdlib>
// experiment switches read from the environment exist in the diagnostics build only; the product build has the default compiled in
static inline int isr_diag_env_int(const char* name, int def)
{
#ifdef ISR_DIAG
    const char* v = getenv(name);
    return v ? atoi(v) : def;
#else
    (void)name;
    return def;
#endif
}
