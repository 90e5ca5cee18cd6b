// hda_krylov.h -- device-resident Krylov solvers (PCG, GMRES) over a local CSR block.
// Parameter contracts: PCG_args src/internal/pcg.c:15-25, GMRES_args src/internal/gmres.c:16-27.
#pragma once

#include "hda_amg.h"

#include <functional>

namespace hda {

struct KrylovParams {
   int    max_iter    = 100;
   double rtol        = 1.0e-6;
   double atol        = 0.0;
   int    two_norm    = 1;
   int    print_level = 0;
   int    krylov_dim  = 30; // GMRES
   int    min_iter    = 0;  // GMRES
   int    skip_real_res_check = 0;
   bool   profile_k1 = false; // bracket each PCG SpMV launch with HIP events
};

struct KrylovResult {
   int                 iters     = 0;
   bool                converged = false;
   double              final_rel = 0.0; // recurrence ||r||/||b|| (HYPRE_*GetFinalRelativeResidualNorm)
   std::vector<double> hist;            // ||r_k||_2, k = 0..iters
   double              k1_ms_sum = 0.0; // profile_k1: summed duration of the PCG SpMV launches
   int                 k1_count  = 0;
   int                 precond_calls = 0; // preconditioner applications actually enqueued (PCG)
};

// Preconditioner seam = what hypre's Krylov expects of a preconditioner
// (HYPRE_Int (*)(void*, void*, void*, void*), src/internal/solver.c:27,94-97):
// z = M^-1 r starting from z = 0; if dot_slot >= 0 the callee must also leave block
// partials of <r, z> in that slot (lets the V-cycle's last sweep fuse PCG's <r, z>).
using PrecondFn = std::function<void(const double *r, double *z, int dot_slot)>;

// The operator a Krylov solver sees: this rank's row block, the plan that refreshes the
// ghost tail of its input vector (null on one rank) and the length work vectors need
// (>= A.ncols and >= the preconditioner's level-0 vector length).
struct LinOp {
   const DCsr     *A      = nullptr;
   const HaloPlan *halo   = nullptr;
   size_t          veclen = 0;
   LinOp() = default;
   LinOp(const DCsr &a) : A(&a), veclen((size_t)std::max(a.ncols, a.nrows)) {}
   LinOp(const DCsr &a, const HaloPlan *h, size_t len) : A(&a), halo(h), veclen(std::max(len, (size_t)std::max(a.ncols, a.nrows))) {}
};

// hypre_PCGSolve (reached from solver_ops[SOLVER_PCG].solve, src/internal/solver.c:211)
KrylovResult pcg(const LinOp &op, const PrecondFn &M, const KrylovParams &p, const double *b, double *x);
// hypre_GMRESSolve (solver_ops[SOLVER_GMRES], src/internal/solver.c:217-228)
KrylovResult gmres(const LinOp &op, const PrecondFn &M, const KrylovParams &p, const double *b, double *x);
// hypre_FlexGMRESSolve (solver_ops[SOLVER_FGMRES], solver.c:229-240) and hypre_BiCGSTABSolve (solver.c:241-252)
KrylovResult fgmres(const LinOp &op, const PrecondFn &M, const KrylovParams &p, const double *b, double *x);
KrylovResult bicgstab(const LinOp &op, const PrecondFn &M, const KrylovParams &p, const double *b, double *x);

// algorithmic HBM bytes of one PCG iteration excluding the preconditioner
int    last_precond_calls(); // preconditioner applications enqueued by the last pcg() call
double pcg_iteration_bytes(const DCsr &A, bool format = false); // format: what the kernels read (coded operators), else CSR figure

} // namespace hda
