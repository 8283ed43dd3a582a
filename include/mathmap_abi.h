/*
 * mathmap_abi.h -- layout-compatible declarations of the reference structures that
 * cross the backend boundary (LP64, non-OPENSTEP build, i.e. what `make` of the
 * reference produces on Linux).  Only data layout is restated: field order, types and
 * sizes follow the cited definitions so that a pointer handed over by the reference's
 * own code can be read here.  Fields this backend never touches keep their size through
 * opaque pointers.  Every struct carries the `mmabi_` prefix; the comment gives the
 * reference type it mirrors.
 */
#ifndef MATHMAP_ABI_H
#define MATHMAP_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned int mmabi_color_t;                 /* color.h:26 */

/* ---- pools (lispreader/pools.h:28-37, mmpools.h:30-39) ---- */
typedef struct { int active_pool; size_t fill_ptr; long *pools[20]; } mmabi_pools_t;
typedef struct { int is_global; mmabi_pools_t pools; void *chunks; } mmabi_mathmap_pools_t;

/* ---- user values (userval.h:52-126) ---- */
enum { MMABI_USERVAL_INT_CONST = 1, MMABI_USERVAL_FLOAT_CONST = 2, MMABI_USERVAL_BOOL_CONST = 3,
       MMABI_USERVAL_COLOR = 4, MMABI_USERVAL_CURVE = 5, MMABI_USERVAL_GRADIENT = 6, MMABI_USERVAL_IMAGE = 7 };

typedef struct mmabi_userval_info {                 /* userval_info_t */
    char *name;
    int type;
    int index;
    union {
        struct { int min, max, default_value; } int_const;
        struct { float min, max, default_value; } float_const;
        struct { int default_value; } bool_const;
        struct { unsigned int flags; } image;
    } v;
    struct mmabi_userval_info *next;
} mmabi_userval_info_t;

typedef struct { int num_control_points; double *control_xs; double *control_ys; float *values; } mmabi_curve_t;
typedef struct { mmabi_color_t *values; } mmabi_gradient_t;
typedef struct { double r, g, b, a; } mmabi_GimpRGB; /* new_template.c.in:89-92 */

struct mmabi_image;
typedef struct mmabi_userval {                      /* userval_t */
    union {
        int int_const;
        float float_const;
        int bool_const;
        struct mmabi_image *image;
        mmabi_curve_t *curve;
        mmabi_gradient_t *gradient;
        struct { mmabi_GimpRGB button_value; mmabi_color_t value; } color;
    } v;
    void *widget_object;                            /* GtkObject* */
} mmabi_userval_t;

/* ---- images (drawable.h:40-174) ---- */
enum { MMABI_IMAGE_DRAWABLE = 1, MMABI_IMAGE_CLOSURE = 2, MMABI_IMAGE_FLOATMAP = 3, MMABI_IMAGE_RESIZE = 4 };

struct mmabi_frame;
struct mmabi_slice;
struct mmabi_invocation;
struct mmabi_input_drawable;

typedef void (*mmabi_init_frame_func_t)(struct mmabi_frame *, struct mmabi_image *);
typedef void (*mmabi_init_slice_func_t)(struct mmabi_slice *, struct mmabi_image *);
typedef void (*mmabi_calc_lines_func_t)(struct mmabi_slice *, struct mmabi_image *, int, int, void *, int);
typedef float *(*mmabi_filter_func_t)(struct mmabi_invocation *, struct mmabi_image *, float, float, float,
                                      mmabi_mathmap_pools_t *);

typedef struct mmabi_mathfuncs {                    /* mathfuncs_t, compiler.h:52-63 */
    mmabi_init_frame_func_t init_frame;
    mmabi_init_slice_func_t init_slice;
    mmabi_calc_lines_func_t calc_lines;
    void *llvm_init_frame_func;
    void *main_filter_func;
    void *init_x_func;
    void *init_y_func;
} mmabi_mathfuncs_t;

typedef struct mmabi_image {                        /* image_t */
    int type;
    int id;
    int pixel_width;
    int pixel_height;
    union {
        struct mmabi_input_drawable *drawable;
        struct {
            mmabi_mathfuncs_t *funcs;
            mmabi_filter_func_t func;
            mmabi_mathmap_pools_t *pools;
            void *xy_vars;
            int num_args;
            mmabi_userval_t args[];
        } closure;
        struct { float ax, bx, ay, by; float *data; } floatmap;
        struct { struct mmabi_image *original; float x_factor, y_factor; } resize;
    } v;
} mmabi_image_t;

typedef struct mmabi_input_drawable {               /* input_drawable_t */
    int used;
    int kind;
    float scale_x, scale_y, middle_x, middle_y;
    union {
        struct {
            void *drawable; int has_selection; int x0, y0; int bpp; int row; int col; void *tile;
            int fast_image_source_width; int fast_image_source_height; mmabi_color_t *fast_image_source;
        } gimp;
        struct { int num_frames; void **cache_entries; char *image_filename; } cmdline;
    } v;
    mmabi_image_t image;
} mmabi_input_drawable_t;

/* ---- filters / mathmap (mathmap.h:59-115) ---- */
enum { MMABI_FILTER_MATHMAP = 1, MMABI_FILTER_NATIVE = 2 };

typedef struct { int row, column, pos; } mmabi_scanner_location_t;
typedef struct { mmabi_scanner_location_t start, end; } mmabi_scanner_region_t;
typedef struct mmabi_option { char *name; struct mmabi_option *suboptions; struct mmabi_option *next; } mmabi_option_t;

typedef struct {                                    /* top_level_decl_t, exprtree.h:226-241 */
    int type;
    char *name;
    char *docstring;
    mmabi_scanner_region_t region;
    union { struct { void *args; mmabi_option_t *options; void *body; } filter; } v;
} mmabi_top_level_decl_t;

typedef struct mmabi_internal {                     /* internal_t, internals.h:35-45 */
    char name[64];
    int index;
    int const_type;
    unsigned int is_used;
    struct mmabi_internal *next;
} mmabi_internal_t;

typedef struct mmabi_variable {                     /* variable_t, vars.h:33-43 */
    char *name;
    struct { int number; int length; } type;
    int index;
    void **compvar;
    int *last_index;
    struct mmabi_variable *next;
} mmabi_variable_t;

typedef struct mmabi_filter {                       /* filter_t */
    int kind;
    char *name;
    int num_uservals;
    mmabi_userval_info_t *userval_infos;
    union {
        struct { mmabi_internal_t *internals; mmabi_variable_t *variables; mmabi_top_level_decl_t *decl; } mathmap;
        struct { int needs_rendered_images; int is_pure; char *func_name; void *func; } native;
    } v;
    struct mmabi_filter *next;
} mmabi_filter_t;

typedef mmabi_mathfuncs_t (*mmabi_initfunc_t)(struct mmabi_invocation *);

typedef struct mmabi_mathmap {                      /* mathmap_t */
    mmabi_filter_t *filters;
    mmabi_filter_t *current_filter;
    mmabi_filter_t *main_filter;
    unsigned int flags;
    mmabi_initfunc_t initfunc;
    mmabi_mathfuncs_t *mathfuncs;
    void *module_info;
    struct mmabi_mathmap *next;
} mmabi_mathmap_t;

/* ---- invocation / frame / slice (mathmap.h:162-226) ---- */
typedef struct mmabi_invocation {                   /* mathmap_invocation_t */
    mmabi_mathmap_t *mathmap;
    mmabi_userval_t *uservals;
    int antialiasing;
    void *orig_val_func;
    int supersampling;
    int output_bpp;
    int edge_behaviour_x, edge_behaviour_y;
    mmabi_color_t edge_color_x, edge_color_y;
    int img_width, img_height;
    int render_width, render_height;
    float image_R;
    int row_stride;
    unsigned char *volatile rows_finished;
    mmabi_mathmap_pools_t pools;
    void *native_filter_cache_mutex;
    void *native_filter_cache_cond;
    void *native_filter_cache;
    mmabi_mathfuncs_t mathfuncs;
    int do_debug;
    int num_debug_tuples;
    void *debug_tuples[8];
} mmabi_invocation_t;

typedef struct mmabi_frame {                        /* mathmap_frame_t */
    mmabi_invocation_t *invocation;
    int frame_render_width, frame_render_height;
    int current_frame;
    float current_t;
    void *xy_vars;
    mmabi_mathmap_pools_t pools;
} mmabi_frame_t;

typedef struct mmabi_slice {                        /* mathmap_slice_t */
    mmabi_frame_t *frame;
    float sampling_offset_x, sampling_offset_y;
    int region_x, region_y, region_width, region_height;
    void *y_vars;
    mmabi_mathmap_pools_t pools;
} mmabi_slice_t;

/* ---- compiler IR (compiler-internals.h:43-235, compiler.h:45-48,70) ---- */
#define MMABI_MAX_OP_ARGS 9
enum { MMABI_TYPE_NIL = 0, MMABI_TYPE_INT, MMABI_TYPE_FLOAT, MMABI_TYPE_COMPLEX, MMABI_TYPE_COLOR, MMABI_TYPE_CURVE,
       MMABI_TYPE_GRADIENT, MMABI_TYPE_IMAGE, MMABI_TYPE_TUPLE, MMABI_TYPE_TREE_VECTOR };   /* ops.lisp:37-66 */

typedef union {                                     /* runtime_value_t (RUNTIME_VALUE_DECL, ops.lisp:367-376) */
    int int_value;
    float float_value;
    float complex_value[2];
    mmabi_color_t color_value;
    void *curve_value, *gradient_value, *image_value, *tuple_value, *tree_vector_value;
} mmabi_runtime_value_t;

typedef struct { int number; int last_index; } mmabi_temporary_t;

struct mmabi_value;
struct mmabi_statement;
typedef struct mmabi_compvar {                      /* compvar_t */
    int index;
    mmabi_variable_t *var;
    mmabi_temporary_t *temp;
    int n;
    int type;
    struct mmabi_value *current;
    struct mmabi_value *values;
} mmabi_compvar_t;

typedef struct mmabi_value {                        /* value_t */
    mmabi_compvar_t *compvar;
    int global_index;
    int index;
    struct mmabi_statement *def;
    void *uses;
    unsigned int const_type : 3;
    unsigned int least_const_type_directly_used_in : 3;
    unsigned int least_const_type_multiply_used_in : 3;
    unsigned int have_defined : 1;
    struct mmabi_value *next;
} mmabi_value_t;

enum { MMABI_PRIMARY_VALUE = 1, MMABI_PRIMARY_CONST = 2 };
typedef struct {                                    /* primary_t */
    int kind;
    int const_type;
    union { mmabi_value_t *value; mmabi_runtime_value_t constant; } v;
} mmabi_primary_t;

typedef struct {                                    /* operation_t */
    int index;
    char *name;
    int num_args;
    int type_prop;
    int is_pure;
    int is_foldable;
    int const_type;
    int arg_types[MMABI_MAX_OP_ARGS];
} mmabi_operation_t;

enum { MMABI_RHS_PRIMARY = 1, MMABI_RHS_INTERNAL, MMABI_RHS_OP, MMABI_RHS_FILTER, MMABI_RHS_CLOSURE, MMABI_RHS_TUPLE,
       MMABI_RHS_TREE_VECTOR };
typedef struct {                                    /* rhs_t */
    int kind;
    union {
        mmabi_primary_t primary;
        mmabi_internal_t *internal;
        struct { mmabi_operation_t *op; mmabi_primary_t args[MMABI_MAX_OP_ARGS]; } op;
        struct { mmabi_filter_t *filter; mmabi_primary_t *args; void *history; } filter;
        struct { mmabi_filter_t *filter; mmabi_primary_t *args; void *history; } closure;
        struct { int length; mmabi_primary_t *args; } tuple;
    } v;
} mmabi_rhs_t;

enum { MMABI_STMT_NIL = 0, MMABI_STMT_ASSIGN, MMABI_STMT_PHI_ASSIGN, MMABI_STMT_IF_COND, MMABI_STMT_WHILE_LOOP };
typedef struct mmabi_statement {                    /* statement_t */
    int kind;
    union {
        struct { mmabi_value_t *lhs; mmabi_rhs_t *rhs; mmabi_rhs_t *rhs2; mmabi_value_t *old_value; } assign;
        struct { mmabi_rhs_t *condition; struct mmabi_statement *consequent, *alternative, *exit; } if_cond;
        struct { struct mmabi_statement *entry; mmabi_rhs_t *invariant; struct mmabi_statement *body; } while_loop;
    } v;
    struct mmabi_statement *parent;
    unsigned int slice_flags;
    struct mmabi_statement *next;
} mmabi_statement_t;

typedef struct { mmabi_filter_t *filter; mmabi_statement_t *first_stmt; } mmabi_filter_code_t;   /* filter_code_t */

#ifdef __cplusplus
}
#endif
#endif
