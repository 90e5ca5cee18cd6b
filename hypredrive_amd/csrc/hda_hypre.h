// hda_hypre.h -- the objects behind the HYPRE_* handles declared in include/HYPRE.h.
#pragma once

#include "../../include/HYPRE.h"
#include "hda_krylov.h"

#include <memory>
#include <string>

struct hypre_IJVector_struct {
   MPI_Comm             comm = MPI_COMM_WORLD;
   long long            jlower = 0, jupper = -1;
   int                  nloc   = 0;
   bool                 initialized = false, assembled = false;
   std::vector<double>  stage;       // host values until Assemble
   hda::DArray<double>  d;           // owned entries (+ optional ghost room) in HBM
   size_t               capacity = 0; // doubles available behind data()
   double              *view = nullptr; // non-owning alias of somebody else's HBM buffer
   std::vector<double>  host_mirror; // backing store for "get values" pointers
   double              *data() { return view ? view : d.data(); }
   const double        *data() const { return view ? view : d.data(); }
   void                 ensure_device();
};

struct hypre_IJMatrix_struct {
   MPI_Comm  comm = MPI_COMM_WORLD;
   long long ilower = 0, iupper = -1, jlower = 0, jupper = -1;
   int       nloc = 0;
   bool      initialized = false, assembled = false;
   // host staging (HYPRE_IJMatrixSetValues / AddToValues before Assemble)
   std::vector<int>       t_row;
   std::vector<long long> t_col;
   std::vector<double>    t_val;
   std::vector<char>      t_add;
   // assembled, HBM resident
   hda::DCsr              A;          // columns: [0,nloc) owned, >= nloc ghosts
   std::vector<long long> ghost_gids; // ascending global ids of the ghost columns
   std::vector<long long> part;       // row starts of every rank (size+1)
   hda::HaloPlan          halo;
   long long              global_rows = 0, global_nnz = 0;
   void                   assemble();
   // adopt a block that was built directly in HBM with global column ids (generator path)
   // (refuse_duplicates: returns false -- nothing adopted -- when a row names a column twice; the caller then stages triplets instead)
   bool adopt_device(int nloc, int nnz, hda::DArray<int> &rowptr, hda::DArray<long long> &gcols, hda::DArray<double> &vals,
                     bool refuse_duplicates = false);
   // a host CSR block with global column ids (HYPREDRV_LinearSystemSetMatrixFromCSR): uploaded as it is, columns mapped, rows sorted
   // and checked on the device; false = a row has a duplicate column (the staged path defines what that means)
   bool assemble_csr(const long long *indptr, const long long *cols, const double *data);
};

enum hda_solver_kind { HDA_SOLVER_PCG = 1, HDA_SOLVER_GMRES = 2, HDA_SOLVER_AMG = 3, HDA_SOLVER_ILU = 4, HDA_SOLVER_FGMRES = 5, HDA_SOLVER_BICGSTAB = 6, HDA_SOLVER_MGR = 7 };

namespace hda {
// addresses of the live solver objects THIS library created: an opaque HYPRE_Solver a caller installs with
// HYPRE_PCGSetPrecond (hypredrive's cookie, a user's own struct) may only be looked into when it is one of them
void solver_registry(const void *p, int op); // op: +1 insert, -1 erase
bool is_live_solver(const void *p);
} // namespace hda

struct hypre_Solver_struct {
   hypre_Solver_struct() { hda::solver_registry(this, +1); }
   ~hypre_Solver_struct() { hda::solver_registry(this, -1); }
   hypre_Solver_struct(const hypre_Solver_struct &) = delete;
   hypre_Solver_struct &operator=(const hypre_Solver_struct &) = delete;
   int                       kind = 0;
   hda::KrylovParams         kp;
   hda::AmgParams            ap;
   // values of setters whose feature is not implemented (checked at Setup)
   const int                *dof_func_ptr = nullptr; // HYPRE_BoomerAMGSetDofFunc (borrowed until Setup)
   double                    agg_trunc[4] = {0.0, 0.0, 0.0, 0.0}; // AggTruncFactor, AggP12TruncFactor, AggPMaxElmts, AggP12MaxElmts: [0] / [2] go to the multipass rows' truncation; the P12 pair belongs to the two-stage interpolation types, which are refused by name
   int                       smooth_type = 5, smooth_num_levels = 0, agg_num_levels = 0, num_functions = 1, cycle_type = 1,
                             restriction = 0, relax_order = 0, sabs = 0, seq_threshold = 0, relax_type_all = -1, filter_functions = 0;
   HYPRE_PtrToSolverFcn      precond = nullptr, precond_setup = nullptr;
   HYPRE_Solver              precond_solver = nullptr;
   std::unique_ptr<hda::Amg> amg;
   // HYPRE_ILU* handle, and the ILU arguments of BoomerAMG's complex smoother (HYPRE_BoomerAMGSetILU*)
   std::unique_ptr<hda::Ilu> ilu;
   hda::IluParams            ilup;
   int                       ilu_type = 0, ilu_fill = 0, ilu_reordering = 0; // checked at Setup: bj-iluk / 0 / 0 only
   hda::DArray<double>       ilu_r, ilu_c;
   // HYPRE_MGR* handle: what the setters recorded (hypre's per-level arrays, copied), built at Setup
   std::unique_ptr<hda::Mgr>     mgr;
   int                           mgr_block_size = 0, mgr_levels = 0, mgr_max_iter = 1, mgr_cycle = 1, mgr_frelax_cycle = 1, mgr_gsmooth_cycle = 1;
   double                        mgr_coarse_th = 0.0;
   std::vector<std::vector<int>> mgr_c_labels; // C labels of every reduction level
   const HYPRE_Int              *mgr_marker = nullptr; // borrowed until Setup, as in hypre
   std::vector<int>              mgr_frelax, mgr_fsweeps, mgr_interp, mgr_restrict, mgr_coarse_method, mgr_gsmooth, mgr_giters;
   HYPRE_Solver                  mgr_csolver = nullptr;
   std::vector<HYPRE_Solver>     mgr_fsolver, mgr_gsolver; // F-relaxation / global smoother handles by level (HYPRE_MGRSetFSolverAtLevel / SetGlobalSmootherAtLevel)
   hda::KrylovResult         last;
   int                       amg_iters = 0;
   double                    amg_rel   = 0.0;
};

namespace hda {
// what a HYPREDRV object holds (defined in hda_hypredrv.hip): level-0 operator, its halo plan, device rhs, hierarchy
bool hypredrv_peek(void *hypredrv, const DCsr **A, const HaloPlan **halo, const double **rhs, Amg **amg);
// error text of the last failed HYPRE_* call (HYPRE_GetError returns the code)
const std::string &hypre_last_error();
int                hypre_set_error(int code, const std::string &msg);
// hints the Krylov loops pass to HYPRE_BoomerAMGSolve through the C callback seam
struct PrecondHints {
   bool zero_guess = false; // the output vector is known to be zero
   int  dot_slot   = -1;    // fuse block partials of <b, x> into the last sweep
   bool dot_done   = false; // set by the callee when it honoured dot_slot
};
PrecondHints &precond_hints();
} // namespace hda
