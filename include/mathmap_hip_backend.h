/*
 * mathmap_hip_backend.h -- the drop-in backend boundary (reference-ABI tier).
 *
 * These are the entry points a MathMap build binds instead of the cc backend's.
 * Each one cites the reference interface it replaces; struct layouts are in
 * mathmap_abi.h.  See INTEGRATION.md for the three-line patch to
 * mathmap_common.c that selects this backend.
 */
#ifndef MATHMAP_HIP_BACKEND_H
#define MATHMAP_HIP_BACKEND_H

#include "mathmap_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces gen_and_load_c_code (compiler.h:79-81, backends/cc.c:634-758).
 * Walks filter_codes[] (indexed like mathmap->filters, native filters' slots unused,
 * compiler.c:4818-4829), lowers the main filter's SSA IR to a HIP kernel string,
 * JIT-compiles it with hiprtc for gfx950 and returns the init function.  On failure
 * returns 0 and leaves the message in the host's `error_string` (exprtree.c:40) when
 * that symbol is visible, and in mmhip_last_error().  `template_filename` and
 * `include_path` are accepted for signature compatibility and ignored (there is no C
 * template).  *module_info receives the handle unload_hip_code() frees. */
mmabi_initfunc_t gen_and_load_hip_code(mmabi_mathmap_t *mathmap, void **module_info, char *template_filename,
                                       char *include_path, mmabi_filter_code_t **filter_codes);

/* Replaces unload_c_code (compiler.h:82, backends/cc.c:760-779). */
void unload_hip_code(void *module_info);

/* The reference's input pixels live behind mathmap_get_pixel (mathmap.h:310,
 * mathmap.c:1309-1319: GIMP tiles or the CLI's image cache).  The backend calls it once
 * per texel when an input drawable is first used, to build the HBM-resident copy.  By
 * default the symbol is looked up in the host process with dlsym(); a host that links
 * statically can hand the function over explicitly. */
typedef mmabi_color_t (*mmabi_get_pixel_func_t)(mmabi_invocation_t *, mmabi_input_drawable_t *, int frame, int x, int y);
void mathmap_hip_set_get_pixel(mmabi_get_pixel_func_t fn);

/* Drops the HBM copy of an input drawable (call when its pixels changed). */
void mathmap_hip_invalidate_drawable(mmabi_input_drawable_t *drawable);

/* Drops the device-side state kept for a host invocation (stream, buffers, native-filter memo).
 * Optional hook for free_invocation (mathmap_common.c:303-319): without it the backend keeps at
 * most a handful of invocations per filter and recycles the rest. */
void mathmap_hip_release_invocation(mmabi_invocation_t *invocation);

#ifdef __cplusplus
}
#endif
#endif
