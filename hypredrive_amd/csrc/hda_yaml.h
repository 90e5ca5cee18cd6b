// hda_yaml.h -- the YAML subset hypredrive's inputs use (reference grammar: SURVEY.md App. B,
// src/internal/yaml.c) and the argument structures it fills (src/internal/args.c:30-45,
// src/internal/pcg.c:15-25, src/internal/gmres.c:16-27, src/internal/amg.c:23-90,120-238).
#pragma once
#include <cstdlib>
#include <cstring>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace hda {

// error bits of the reference's include/internal/error.h:16-48 (part of the ABI: the
// uint32_t every HYPREDRV_* call returns is an OR of these)
enum : uint32_t {
   ERR_NONE = 0x0, ERR_YAML_INVALID_INDENT = 0x1, ERR_YAML_INVALID_BASE_INDENT = 0x2,
   ERR_YAML_INCONSISTENT_INDENT = 0x4, ERR_YAML_INVALID_DIVISOR = 0x8, ERR_YAML_TREE_NULL = 0x10,
   ERR_YAML_TREE_INVALID = 0x20, ERR_YAML_MIXED_INDENT = 0x40, ERR_YAML_INVALID_INDENT_JUMP = 0x80,
   ERR_INVALID_KEY = 0x100, ERR_INVALID_VAL = 0x200, ERR_UNEXPECTED_VAL = 0x400, ERR_MAYBE_INVALID_VAL = 0x800,
   ERR_MISSING_KEY = 0x1000, ERR_EXTRA_KEY = 0x2000, ERR_MISSING_SOLVER = 0x4000, ERR_MISSING_PRECON = 0x8000,
   ERR_MISSING_DOFMAP = 0x10000, ERR_INVALID_SOLVER = 0x20000, ERR_INVALID_PRECON = 0x40000,
   ERR_FILE_NOT_FOUND = 0x80000, ERR_FILE_UNEXPECTED_ENTRY = 0x100000, ERR_UNKNOWN_HYPREDRV_OBJ = 0x200000,
   ERR_HYPREDRV_NOT_INITIALIZED = 0x400000, ERR_UNKNOWN_TIMING = 0x800000, ERR_HYPRE_INTERNAL = 0x1000000,
   ERR_MISSING_LIB = 0x2000000, ERR_ALLOCATION = 0x20000000, ERR_OUT_OF_BOUNDS = 0x40000000, ERR_UNKNOWN = 0x80000000u
};

struct YNode {
   std::string                         key, val; // val empty for pure containers
   bool                                seq_item = false;
   std::vector<std::unique_ptr<YNode>> kids;
   YNode *find(const std::string &k);
   YNode *get_or_add(const std::string &k);
};

// Parses text into root->kids. Returns error bits; message describes the first problem.
uint32_t yaml_parse(const std::string &text, YNode &root, std::string &message);
void     yaml_print(const YNode &root, FILE *out);
// "--a:b:c value" override (reference: src/internal/yaml.c:2178)
void     yaml_override(YNode &root, const std::string &path, const std::string &value);
// "include: other.yml" anywhere in the tree is replaced by the parsed content of that file, looked up relative to
// base_dir (reference src/internal/yaml.c:28-235: nested includes, depth limit, cycle detection)
uint32_t yaml_expand_includes(YNode &root, const std::string &base_dir, std::string &message);

struct GeneralArgs {
   std::string name, statistics_filename;
   int         warmup = 0, statistics = 1, print_config_params = 1, use_millisec = 0, num_repetitions = 1, exec_policy = 1;
};
struct LSArgs {
   std::string dirname, matrix_filename, matrix_basename, precmat_filename, precmat_basename, rhs_filename, rhs_basename, x0_filename,
      xref_filename, dofmap_filename, dofmap_basename, sol_filename;
   std::vector<int> set_suffix; // explicit suffixes of systems 1, 2, ... (reference linsys.c:790-805)
   int digits_suffix = 5, init_suffix = -1, last_suffix = -1, init_guess_mode = 0, rhs_mode = 2, type = 1, num_systems = 1;
};
// solver.scaling (Scaling_args, reference include/internal/scaling.h:32-37, defaults src/internal/scaling.c:71-76)
struct ScalingArgs {
   int                 enabled = 0;
   int                 type    = 0; // scaling_type_t order: rhs_l2, dofmap_mag, dofmap_custom, dofmap_row_custom, dofmap_col_custom, dofmap_similarity_custom
   std::vector<double> custom_values;
};
struct KrylovArgs {
   ScalingArgs scaling;
   int    method = 0; // 0 pcg, 1 gmres, 2 fgmres, 3 bicgstab (reference solver_t order)
   // PCG_args / GMRES_args union
   int    max_iter = 100, two_norm = 1, stop_crit = 0, rel_change = 0, print_level = 1, recompute_res = 0;
   double relative_tol = 1.0e-6, absolute_tol = 0.0, residual_tol = 0.0, conv_fac_tol = 0.0;
   int    min_iter = 0, skip_real_res_check = 0, krylov_dim = 30, logging = 1;
   void   defaults_for(int m);
};
struct IluArgs { // ILU_args, defaults of src/internal/ilu.c:15-28
   int    max_iter = 1, print_level = 0, type = 0, fill_level = 0, reordering = 0, tri_solve = 1, lower_jac_iters = 5,
          upper_jac_iters = 5, max_row_nnz = 200, schur_max_iter = 3;
   double droptol = 1.0e-2, nsh_droptol = 1.0e-2, tolerance = 0.0;
};
struct AmgArgs { // AMG_args, GPU-branch defaults of src/internal/amg.c:120-238
   int    max_iter = 1, print_level = 0;
   double tolerance = 0.0;
   // interpolation
   int    prolongation_type = 6, restriction_type = 0, max_nnz_row = 4;
   double trunc_factor = 0.0, restrict_strong_th = 0.25, restrict_filter_th = 0.0;
   // coarsening
   int    type = 8, rap2 = 0, mod_rap2 = 1, keep_transpose = 1, sabs = 0, num_functions = 1, filter_functions = 0, nodal = 0,
          seq_amg_th = 0, min_coarse_size = 0, max_coarse_size = 64, max_levels = 25;
   double max_row_sum = 0.9, strong_th = 0.25;
   // aggressive
   int    agg_num_levels = 0, agg_num_paths = 1, agg_prolongation_type = 4, agg_max_nnz_row = 0;
   double agg_trunc_factor = 0.0, agg_P12_max_elements = 0.0, agg_P12_trunc_factor = 0.0;
   // relaxation
   int    relax_type = -1, down_type = 18, up_type = 18, coarse_type = 9, down_sweeps = -1, up_sweeps = -1, coarse_sweeps = 1,
          num_sweeps = 1, order = 0, points = 0;
   double weight = 1.0, outer_weight = 1.0;
   // relaxation.chebyshev (Cheby_args, reference src/internal/cheby.c:15-20)
   int    cheby_order = 2, cheby_eig_est = 10, cheby_variant = 0, cheby_scale = 1;
   double cheby_fraction = 0.3;
   // complex smoother
   int    smooth_type = 5, smooth_num_levels = 0, smooth_num_sweeps = 1;
   IluArgs smooth_ilu;
   // The reference picks these defaults at COMPILE time (#ifdef HYPRE_USING_GPU, amg.c:138-146,
   // 183-189); this library is a GPU build.  HYPREDRV_AMD_DEFAULTS=cpu makes it start from the
   // defaults of a CPU build instead (HMIS, hybrid l1 Gauss-Seidel 13/14, no mod_rap2 /
   // keep_transpose): the settings the reference's checked-in outputs were produced with, for
   // drivers that leave no other way to select them.
   AmgArgs()
   {
      const char *e = getenv("HYPREDRV_AMD_DEFAULTS");
      if (e && !strcmp(e, "cpu")) { type = 10; down_type = 13; up_type = 14; mod_rap2 = 0; keep_transpose = 0; }
   }
};
// NestedKrylov_args (reference include/internal/krylov.h, src/internal/krylov.c:336-414): a Krylov solver named inside an MGR
// component, with an optional preconditioner block of its own
struct NestedKrylovArgs {
   bool       set = false;
   KrylovArgs solver;
   int        precon = 99; // 99 none, 0 amg, 2 ilu
   AmgArgs    amg;
   IluArgs    ilu;
};
// MGR_args / MGRlvl_args (reference include/internal/mgr.h:132-178; defaults src/internal/mgr.c:1226-1330)
struct MgrLevelArgs {
   std::vector<int> f_dofs;
   int prolongation_type = 0, restriction_type = 0, coarse_level_type = 0;
   int f_type = 7, f_sweeps = 1; // f_relaxation
   int g_type = -1, g_sweeps = 1; // g_relaxation
   std::string f_block, g_block;  // nested solver blocks named here (only f_relaxation.amg is implemented)
   AmgArgs     f_amg;             // f_relaxation: {amg: {...}}
   IluArgs     f_ilu, g_ilu;      // f_relaxation / g_relaxation: {ilu: {...}}
   bool        g_ilu_block = false; // g_relaxation came with its own ilu block (a smoother object, not hypre's built-in type 16)
   NestedKrylovArgs f_krylov;     // f_relaxation: {gmres: {...}}
};
struct MgrArgs {
   int    non_c_to_f = 1, pmax = 0, max_iter = 1, num_levels = 0, relax_type = 7, print_level = 0, nonglk_max_elmts = 1, cycle = 1, cycle_smooth_pos = 1; // cycle_smooth_pos: 1 pre, 2 post, 3 pre + post (mgr.h:160)
   double tolerance = 0.0, coarse_th = 0.0;
   std::vector<MgrLevelArgs> level;
   int         coarsest_type = -1; // -1 / 0: BoomerAMG
   std::string coarsest_block;     // anything but amg is not implemented
   AmgArgs     coarsest_amg;
   IluArgs     coarsest_ilu;
   NestedKrylovArgs coarsest_krylov; // coarsest_level: {gmres: {...}}
};
struct PreconArgs {
   int         method = 0; // 0 boomeramg, 1 mgr, 2 ilu, 3 fsai, ... 99 none
   std::string method_name = "amg";
   AmgArgs     amg;
   IluArgs     ilu;
   MgrArgs     mgr;
};
// preconditioner.reuse (reference src/internal/precon_reuse.c:2280-2567): the static policy
struct ReuseArgs {
   int              enabled = 0, frequency = 0;
   std::vector<int> linear_system_ids; // rebuild exactly on these systems ("always" = {0})
};
struct InputArgs {
   GeneralArgs             general;
   LSArgs                  ls;
   KrylovArgs              solver;
   std::vector<PreconArgs> precon_variants; // >= 1
   int                     active_variant = 0;
   bool                    has_precon = false;
   ReuseArgs               reuse;
   PreconArgs             &precon() { return precon_variants[(size_t)active_variant]; }
};

// Fill args from a parsed tree.  lib_mode: linear_system optional, print_config_params off.
uint32_t args_from_yaml(YNode &root, bool lib_mode, InputArgs &args, std::string &message);
uint32_t precon_from_text(const std::string &yaml_text, PreconArgs &out, std::string &message); // presets
uint32_t solver_from_text(const std::string &yaml_text, KrylovArgs &out, std::string &message);

} // namespace hda
