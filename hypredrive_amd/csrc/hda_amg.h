// hda_amg.h -- device-resident BoomerAMG-style hierarchy and V-cycle (single rank block).
// Parameter contract: reference AMG_args (include/internal/amg.h:108-123) as forwarded by
// hypredrv_AMGCreate (src/internal/amg.c:864-1035) to the HYPRE_BoomerAMGSet* setters.
#pragma once

#include "hda_dist.h"
#include "hda_kernels.h"

#include <memory>
#include <vector>

namespace hda {

// ILU_args subset that is implemented (src/internal/ilu.c:15-28): type bj-iluk, fill_level 0, reordering 0
struct IluParams {
   int tri_solve = 1;              // 1 exact substitutions (level scheduled), 0 Jacobi iterations on L and U
   int lower_it = 5, upper_it = 5; // lower_jac_iters / upper_jac_iters
   int max_iter = 1;               // iterations x += M^-1 (b - A x) per solve
   // V contiguous row blocks on one GPU = bj-iluk at np = V: the factorisation drops the entries that leave a row's block, and
   // the exact substitutions (tri_solve 1) run block-parallel on the block Gauss-Seidel kernels.  1 = one block, 0 = chosen from
   // the operator's size and bandwidth (amg_auto_blocks), V > 1 = hypre's even split unless block_part names the V + 1 row starts
   int                    blocks = 1;
   std::vector<long long> block_part;
};

struct AmgParams {
   // coarsening (src/internal/amg.c:138-157)
   int    coarsen_type    = 8;  // PMIS (hypre-GPU default); 10 HMIS = Ruge first pass (one device thread, small systems); others not on device
   double strong_th       = 0.25;
   double max_row_sum     = 0.9;
   int    max_coarse_size = 64;
   int    min_coarse_size = 0;
   int    max_levels      = 25;
   // interpolation (amg.c:120-128)
   int    interp_type  = 6; // extended+i; 17: mm-ext+i, its matrix-matrix form (an operator of its own, one rank); 3: direct with separation of weights (one rank)
   int    pmax         = 4;
   double trunc_factor = 0.0;
   // relaxation (amg.c:178-199)
   int    relax_down = 18, relax_up = 18, relax_coarse = 9;
   int    sweeps_down = 1, sweeps_up = 1, sweeps_coarse = 1;
   double relax_weight = 1.0, outer_weight = 1.0;
   // solver knobs (amg.c:222-226)
   int    max_iter = 1;
   double tol      = 0.0;
   int    print_level = 0;
   uint64_t seed = 2747; // PMIS tie-break hash seed
   int    num_functions = 1; // coarsening.num_functions; > 1 = systems AMG, unknown approach (presets elasticity_2d/3d)
   // relaxation.chebyshev (relax type 16; reference src/internal/cheby.c:15-20, amg.c:886-890)
   int    cheby_order = 2, cheby_eig_est = 10, cheby_variant = 0, cheby_scale = 1;
   double cheby_fraction = 0.3;
   // complex smoother (amg.c:899-921): type 5 = ILU on levels < smooth_num_levels, replacing the relaxation
   // sweeps there; one smoothing step = smooth_num_sweeps iterations u += M^-1 (f - A u)
   int       smooth_type = 5, smooth_num_levels = 0, smooth_num_sweeps = 1;
   IluParams ilu;
   // aggressive coarsening (AMGagg_args, amg.c:160-173, 938-944; hda_amg_agg.hip): on the first agg_num_levels levels a second PMIS
   // pass over the distance-two strength graph (>= agg_num_paths paths of length <= 2) and multipass interpolation (type 4, untruncated)
   int agg_num_levels = 0, agg_num_paths = 1, agg_interp_type = 4;
   int    agg_pmax = 0;           // aggressive.max_nnz_row (HYPRE_BoomerAMGSetAggPMaxElmts): 0 = no limit
   double agg_trunc_factor = 0.0; // aggressive.trunc_factor (HYPRE_BoomerAMGSetAggTruncFactor)
   // V contiguous row blocks on one GPU = the reference at np = V (its CPU defaults, amg.c:141-146 HMIS and :182-189 hybrid l1
   // Gauss-Seidel, are rank-block algorithms): the hybrid Gauss-Seidel sweeps (3/4/6/8/13/14) are Gauss-Seidel inside a block and
   // Jacobi across blocks with hypre's option-4 l1 divisor, HMIS is a Ruge first pass per block + PMIS on what it leaves; coarse
   // levels inherit the blocks through their C points.  1 = one block (the sequential algorithms), 0 = chosen by the setup from
   // the operator's size and bandwidth (amg_auto_blocks), V > 1 = hypre's even split into V blocks unless block_part names the
   // V + 1 row starts.
   int                    blocks = 1;
   std::vector<long long> block_part;
};
int amg_auto_blocks(const DCsr &A); // the setup's choice for blocks = 0

// dependency levels of the local pattern for Gauss-Seidel sweeps (hda_gs.hip)
struct GsPlan {
   DArray<int>                      perm;      // rows grouped by level, ascending inside a level
   mutable const int               *span_rp = nullptr; // the row pointer array the spans below were read from (a reused hierarchy is
   mutable int                      span_nnz = -1;     // applied with the level-0 matrix of the call: spans follow it, gs_sweep checks)
   mutable unsigned long long       span_gen = 0;      // DCsr::gen of that matrix (address + size alone can repeat after a re-assembly)
   mutable DArray<int>              rbeg, rend; // first / past-the-end entry of the row at every sorted position (rowptr[perm[q]], rowptr[perm[q] + 1]):
                                               // read coalesced with perm, so a sweep does not chase row id -> row pointer -> entries
   DArray<int>                      d_lvl_ptr; // device copy of lvl_ptr
   std::vector<int>                 lvl_ptr;   // nlev + 1 offsets into perm
   std::vector<std::pair<int, int>> segments;  // launch groups [first level, last level)
   int                              nlev  = 0;
   bool                             built = false;
   // row-block form (nblk > 1): Gauss-Seidel inside a block, Jacobi across blocks.  perm lists block after block (block q owns the
   // positions [part[q], part[q + 1])), inside a block dependency level after level of the block's OWN pattern; one workgroup
   // sweeps one block
   int         nblk = 0;
   DArray<int> blk_part;    // nblk + 1 row starts
   DArray<int> blk_lvl_ptr; // nblk + 1: block q's levels are blk_lvl[blk_lvl_ptr[q] .. blk_lvl_ptr[q + 1]]
   DArray<int> blk_lvl;     // first position of every (block, level); one past the end = nrows
   DArray<int> group_of_pos; // exclusive scan of the (block, level) group starts over the positions (n + 1): position q is in group [q + 1] - 1
   int         blk_max_levels = 0;
   double      blk_mean_rows_per_level = 0.0;
   // sweep-order copy of the operator (big levels): the rows in perm order, entries contiguous, a column inside the row's block named
   // by its POSITION in perm, a column outside by ~column -- a level's rows, their entries and (on a grid) their neighbours' values
   // then sit next to each other, where the row-ordered arrays give every row cache lines of its own
   mutable DArray<int>    s_rowptr, s_col;
   mutable DArray<double> s_aii; // diagonal entries in sweep order (the sweep-order copy keeps them apart from the rows' chunks)
   mutable DArray<double> s_val;
   mutable DArray<double> s_x, s_b, s_d; // the sweep's iterate, right-hand side and divisors in sweep order
   // what s_d / s_b hold (round 5): the divisors of a level do not change between sweeps and the right-hand side of a level is the same
   // for every sweep of one cycle, so their sweep-order copies are made once (per divisor array / per cycle) instead of per sweep
   mutable const double  *sd_src = nullptr, *sb_src = nullptr;
   mutable const double  *to_b = nullptr, *to_d = nullptr; // what the next k_gs_to_sweep_order gathers (nullptr: s_b / s_d are current)
   mutable bool           sorted = false;
   // level-wise form of that copy (k_gs_blocks_ring): the rows of one (block, level) all take the level's widest row's number of
   // 4-entry chunks, so a row's chunks sit at first chunk of the level + row * width -- no row pointer to chase; r_cb / r_w per
   // (block, level) in the order of blk_lvl
   mutable DArray<int>    r_cb, r_w, r_col;
   mutable DArray<double> r_val, s_x0; // (s_x0: the iterate at the start of the sweep in sweep order -- what the other blocks' columns read)
   mutable bool           ring = false;
   // the sweep as a list of PASSES per block and direction (8 ints each: first position, rows, first chunk, chunks per row, the
   // positions whose values come from the LDS ring, barrier-after flag), so that the kernel's control flow is one scalar load
   mutable DArray<int>    r_pass[2], r_pass_ptr; // [0] forward, [1] backward; r_pass_ptr: nblk + 1 offsets (in passes, 2 spare ones per block)
   mutable int            ring_lpr = 0, ring_nt = 0;
   mutable int            free_lpr = 0, free_ring = 0, free_maxc = 1; // barrier-free kernel (k_gs_blocks_free): lanes per row (0 = not applicable), LDS ring slots, chunks per lane
   mutable bool           free_long = false;                          // ... and whether rows beyond lanes x chunks exist (kernel with the long-row path)
   // the dependency copy: of every row the in-block columns EARLIER in the sweep alone -- all that a forward sweep from the zero guess needs
   mutable DArray<int>    d_rowptr, d_col;
   mutable DArray<double> d_val;
   mutable int            dep_lpr = 0, dep_ring = 0, dep_maxc = 1; // (0 lanes: no such copy)
   mutable bool           dep_long = false;
   std::vector<int>       h_part;                                    // host copy of the blocks' row starts
   std::vector<int>       h_blk_lvl, h_blk_lvl_ptr; // host copies of blk_lvl / blk_lvl_ptr (the copies are rebuilt when a kept plan meets another matrix)
};
void build_gs_plan(const DCsr &A, GsPlan &plan);
void build_gs_plan_blocks(const DCsr &A, const std::vector<int> &part, GsPlan &plan);
// one hybrid Gauss-Seidel sweep in place: x_i += dinv_i (b_i - A_i x), rows in sequential order
void gs_sweep(const DCsr &A, const GsPlan &plan, const double *dinv, const double *b, double *x, bool forward);
// row-block form: xout = sweep(xin); the other blocks' values are read from xin, which the sweep leaves alone (xin != xout).
// zero_in: the input is the zero vector and is not read (xin may be null or xout)
void gs_free_check(); // throws if a barrier-free block sweep hit its spin limit since the last call (Krylov solves call it at their end)
// b_unchanged: the caller vouches that b (the same array, the same contents) was the right-hand side of the previous sweep on this plan
void gs_sweep_blocks(const DCsr &A, const GsPlan &plan, const double *dinv, const double *b, const double *xin, double *xout,
                     bool forward, bool zero_in, bool b_unchanged = false);

// block-Jacobi ILU(0) of a rank's diagonal block (hda_ilu.hip)
class Ilu {
 public:
   void        setup(const DCsr &A, const IluParams &p); // columns >= A.nrows (ghosts) are dropped
   void        apply(const double *r, double *z);        // z = U^-1 L^-1 r; r and z must not alias
   const DCsr &factors() const { return LU; }            // strict lower part = L (unit diagonal), rest = U
   double      apply_bytes() const;
   IluParams   prm;

 private:
   DCsr           LU;
   DCsr           Ls, Us; // tri_solve 0: strict lower triangle / diagonal + upper triangle, one stream each
   DArray<int>    diag;
   GsPlan         plan;
   DArray<double> work, dinv;
   // row blocks (IluParams::blocks): the substitutions as two zero-guess block sweeps over LU -- forward with unit divisors
   // (the upper part multiplies the zero guess), backward with 1 / u_ii (the lower part does)
   std::vector<int> bpart;
   GsPlan           bplan;
   DArray<double>   ones, udinv;

 public:
   int blocks_used() const { return bpart.empty() ? 1 : (int)bpart.size() - 1; }
};

void ilu_solve(Ilu &F, const DCsr &A, const HaloPlan *halo, const double *b, double *x, bool zero_guess, DArray<double> &r,
               DArray<double> &c); // max_iter iterations x += M^-1 (b - A x)

// PCG <-> preconditioner: a V-cycle from a zero guess that opens with the Jacobi sweep z0 = dinv .* r on level 0 lets its caller do
// that multiplication where r is produced (PCG's update kernel: one pass over r and one launch less per iteration).  Thread-local,
// like the hints of the C callback seam (hda_hypre.h PrecondHints).  The callee publishes an offer after every application
// (Amg::apply_offering); the caller that honours it writes dinv .* r into the offered vector and sets `done` before the next
// application, which skips its first sweep and clears the flag -- a flag still set afterwards means the callee never looked.
struct FirstSweepFusion {
   bool          valid = false;   // offer: the next application from a zero guess may have its first sweep done by the caller
   const double *dinv  = nullptr; //   divisors of that sweep
   double       *dest  = nullptr; //   where the sweep's result is expected; nullptr = the application's output vector itself
   int           n     = 0;       //   rows of the operator the offer is about (a caller iterating on another system must not take it)
   const void   *owner = nullptr; //   the preconditioner that made the offer
   bool          done  = false;   // caller -> callee: it has been done ...
   const double *in    = nullptr; //   ... for the application that gets THIS right-hand side
   double       *out   = nullptr; //   ... and THIS output vector: any other application runs its own sweep
};
FirstSweepFusion &first_sweep_fusion();

struct AmgLevel {
   DCsr           A, P, R;
   // row partitions: the P rows of this rank's GHOST fine points (the ghost slots of hA, in their order; columns as P's), so that
   // the prolongation updates the ghost copies of the iterate too and the first post-smoothing sweep needs no exchange
   DCsr           Pg;
   bool           pg_ready = false; // set by the setup on EVERY rank or on none (the exchange it saves is a collective)
   GsPlan         gs;
   std::unique_ptr<Ilu> ilu; // complex smoother of this level, if any
   // Chebyshev smoother (relax type 16): D^-1/2 scaling, polynomial coefficients, work vectors
   DArray<double> cheb_ds, cheb_v, cheb_w;
   double         cheb_coef[5] = {0, 0, 0, 0, 0}, cheb_max_eig = 0.0, cheb_min_eig = 0.0;
   DArray<double> ilu_r, ilu_c;
   DArray<int>    cf;
   std::vector<int> blk_part; // row blocks of this level (AmgParams::blocks): V + 1 row starts; empty = one block
   DArray<double> dinv_down, dinv_up; // relax_weight / l1 (or / a_ii), per cycle direction
   DArray<double> f, u, u2, t;
   bool           gs_b_seen = false; // a block sweep of this cycle has already put this level's right-hand side into sweep order
   // row-partitioned runs: ghost refresh plans for the inputs of A_l, P_l, R_l and the
   // length every level-l work vector needs ([owned | largest ghost tail])
   HaloPlan hA, hP, hR;
   size_t   ext = 0;
};

class Amg {
 public:
   explicit Amg(const AmgParams &p) : prm(p)
   { // relax type 89 ("l1sym-hgs", reference src/internal/amg.c:375) is hypre's second name for the symmetric hybrid l1
     // Gauss-Seidel sweep (forward, then backward), type 8 here
      for (int *t : {&prm.relax_down, &prm.relax_up, &prm.relax_coarse})
         if (*t == 89) *t = 8;
   }
   // systems AMG (prm.num_functions > 1): function of every level-0 unknown of the matrix handed to
   // setup() (for setup_dist: of this rank's rows); empty = (dof_row_offset + i) mod num_functions
   std::vector<int> dof_func0;
   long long        dof_row_offset = 0;
   // hypre_BoomerAMGSetup (src/internal/precon.c:107): A is borrowed for level 0.
   void setup(const DCsr &A);
   // Row-partitioned variant: Aloc is this rank's block ([owned | ghost] columns, ghosts =
   // ghost_gids ascending), hA0 its halo plan, part0 the row starts of all ranks.  Round-1
   // scheme: the operator is gathered and the hierarchy is built redundantly on every rank
   // (bit-identical to the 1-GPU hierarchy because PMIS weights hash the global row id),
   // then every level is cut into local blocks + halo plans for a distributed V-cycle.
   void setup_dist(const DCsr &Aloc, const HaloPlan &hA0, const std::vector<long long> &part0,
                   const std::vector<long long> &ghost_gids0);
   // fully partitioned setup (distributed PMIS / ext+i / RAP; only levels below HDA_REPLICATE_ROWS are gathered)
   void setup_dist_partitioned(const DCsr &Aloc, const HaloPlan &hA0, const std::vector<long long> &part0,
                               const std::vector<long long> &ghost_gids0);
   // A reused hierarchy applied to a later system of a sequence (preconditioner.reuse): hypre's
   // BoomerAMGSolve takes level 0 from the matrix of the call, everything else from the setup.  Same here:
   // the level-0 operator (and its halo plan) is swapped, divisors and coarse levels stay.
   void rebind(const DCsr &A, const HaloPlan *hA);
   // (a later matrix can land on the address of a freed one: compare the sizes seen at setup / rebind too)
   bool bound_to(const DCsr &A) const { return A0 == &A && A.nrows == a0_dims[0] && A.ncols == a0_dims[1] && A.nnz == a0_dims[2]; }
   // length of the level-0 vectors handed to apply()/solve() (x must have this room)
   // solve-phase renumbering of the coarse levels (hda_reorder.hip); 0 = none
   void   reorder_levels();
   int    reordered_levels = 0;
   size_t vec_len0() const { return levels.empty() ? 0 : levels[0].ext; }
   bool   distributed() const { return dist; }
   // row-partitioned runs: levels whose operator is cut into row blocks (the rest is the replicated tail every rank cycles redundantly)
   int    partitioned_levels() const { return dist ? (int)levels.size() - (tail ? 1 : 0) : 0; }
   int    total_levels() const { return (int)levels.size() + (tail ? tail->num_levels() - 1 : 0); } // partitioned levels + replicated tail
   // HYPRE_BoomerAMGSolve as a preconditioner (precon.c:108): one V(nu1,nu2) from x = 0.
   // dot_slot >= 0: also emit block partials of <b, x> (fuses PCG's <r, z>).
   void apply(const double *b, double *x, int dot_slot = -1);
   // the same, as the preconditioner of one of this library's Krylov loops: honours and renews the first-sweep offer (FirstSweepFusion)
   void apply_offering(const double *b, double *x, int dot_slot = -1);
   bool first_sweep_fusable() const;
   double *first_sweep_dest(double *x); // where apply(b, x) puts its zero-guess first sweep: x or the level's second buffer
   // general solve entry: max_iter cycles starting from the x passed in
   void solve(const double *b, double *x);

   int           num_levels() const { return (int)levels.size(); }
   const DCsr   &level_A(int l) const { return l == 0 ? *A0 : levels[l].A; }
   AmgLevel     &level(int l) { return levels[l]; }
   double        operator_complexity() const;
   double        grid_complexity() const;
   // algorithmic HBM bytes of one V-cycle (SURVEY 8(d) formulas on the built hierarchy)
   double        vcycle_bytes(bool format = false) const; // format: bytes the kernels read (coded operators), else the CSR figure
   AmgParams     prm;
   int           blocks_used = 1;      // row blocks the setup worked with (AmgParams::blocks resolved)
   const std::vector<int> &level_blocks(int l) const { return levels[(size_t)l].blk_part; }
   double        setup_times[8] = {0}; // strength, coarsen, interp, rap, misc (diagnostic)

 private:
   void cycle(const double *b, double *x, bool zero_guess, int dot_slot, bool first_sweep_given = false);
   void relax(int l, int type, const double *dinv, const double *b, double *&cur, double *&alt,
              bool zero_guess, int dot_slot);
   void build_hierarchy(const DCsr &A);
   void build_smoother_data(int l); // divisors (and Gauss-Seidel level sets) of level l on the matrix the cycle uses
   void build_cheby(int l);         // eigenvalue estimate and polynomial of the Chebyshev smoother
   void cheby_sweep(int l, const double *b, double *u, bool zero_guess, bool ghosts_fresh = false);
   bool ghosts_fresh_ = false; // cycle -> relax: the next sweep's input has fresh ghost copies (AmgLevel::Pg)
   void coarse_solve(const double *f, double *u);
   const HaloPlan &level_hA(int l) const { return (l == 0 && hA0) ? *hA0 : levels[l].hA; }
   const DCsr           *A0 = nullptr;
   int                   a0_dims[3] = {0, 0, 0}; // rows, columns, entries of the level-0 operator when it was bound
   const HaloPlan       *hA0 = nullptr;
   bool                  dist = false;
   long long             coarse_lo = 0; // first coarsest row owned by this rank
   int                   coarse_nloc = 0;
   double                stats_nnz[32] = {0}, stats_rows[32] = {0}; // global sizes per level
   int                   stats_levels = 0;
   int                   level0 = 0; // PMIS level index of this hierarchy's level 0 (tail of a partitioned hierarchy)
   DArray<double>        cbuf_f, cbuf_u;
   // row-partitioned runs: levels with few rows stay whole on every rank and are cycled
   // redundantly after one small all-reduce of the restricted residual
   std::unique_ptr<Amg>  tail;
   DCsr                  own_A0; // level-0 operator of a tail hierarchy
   void                  adopt_tail(Amg &parent, int first_level);
   std::vector<AmgLevel> levels;
   DArray<double>        coarse_invT; // dense inverse of the coarsest operator (column-major)
   int                   coarse_n = 0;
   bool                  coarse_dense = false;
};

// row-partitioned sparse product and reverse halo sum (hda_amg_setup.hip), shared by the partitioned setups
void dist_spgemm(const DCsr &X, const HaloPlan &hX, const DCsr &Y, const std::vector<long long> &y_ghosts,
                 const std::vector<long long> &part_c, DCsr &C, std::vector<long long> &c_ghosts);
void halo_reverse_add(const HaloPlan &h, double *x_ext);

// ---- MGR (hda_mgr.hip): multigrid reduction by dof labels, BoomerAMG on the coarsest system.
// Parameter contract: reference MGR_args / MGRlvl_args (include/internal/mgr.h:132-178), defaults src/internal/mgr.c:1226-1330.
struct NestedKrylov { // the subset of KrylovParams a nested solve uses (hda_krylov.h includes this header)
   int    max_iter = 100, krylov_dim = 30, min_iter = 0, two_norm = 1, skip_real_res_check = 0;
   double rtol = 1.0e-6, atol = 0.0;
};
struct MgrLevelParams {
   std::vector<int> f_labels;          // level.N.f_dofs
   int interp_type = 0;                // prolongation_type: 0 injection, 1 l1-jacobi, 2 jacobi
   int restrict_type = 0;              // restriction_type: 0 injection, 2 jacobi, 14 columped
   int coarse_type = 0;                // coarse_level_type: 0 rap
   int frelax_type = 7, frelax_sweeps = 1;   // f_relaxation: 7 jacobi, 18 l1-jacobi, 2 amg (one BoomerAMG cycle on A_FF), 32 ilu (ILU(0) of A_FF)
   AmgParams frelax_amg;                     // f_relaxation.amg block
   IluParams ilu;                            // ILU arguments of this level's ILU components (f_relaxation 32, g_relaxation 16)
   int grelax_type = -1, grelax_sweeps = 1;  // g_relaxation: -1 none, 3/4/6/13/14 hybrid GS, 88 l1-hsgs, 16 ilu
   int grelax_blocks = 1;                    // row blocks of the hybrid GS global relaxation (as AmgParams::blocks: 1 one block, 0 the setup's choice, V even split)
   // f_relaxation: {gmres: {..., preconditioner: {amg: ...}}} -- a nested Krylov solve on A_FF (reference src/internal/krylov.c,
   // mgr.c:3938-3960).  method: -1 none, 0 pcg, 1 gmres, 2 fgmres, 3 bicgstab; its preconditioner is the level's amg / ilu
   // component (frelax_type 2 / 32) or nothing (fkrylov_precond false).
   int          fkrylov_method = -1;
   NestedKrylov fkrylov;
   bool         fkrylov_precond = true;
};
struct MgrParams {
   std::vector<MgrLevelParams> levels;
   AmgParams                   coarse; // coarsest_level: amg
   bool                        coarse_is_ilu = false; // coarsest_level: ilu
   IluParams                   coarse_ilu;
   int                         max_iter = 1;
   int                         cycle = 1;                      // 1 V, 2 W (mgr.c:614-675)
   int                         frelax_pos = 1, gsmooth_pos = 1; // smoothing positions: 1 before the coarse correction, 2 after, 3 both
   // coarsest_level: {gmres: {..., preconditioner: ...}} -- a nested Krylov solve of the coarsest system (mgr.c:4253-4275)
   int                         ckrylov_method = -1;
   NestedKrylov                ckrylov;
   bool                        ckrylov_precond = true;
};
class Mgr {
 public:
   explicit Mgr(const MgrParams &p) : prm(p) {}
   void        setup(const DCsr &A, const std::vector<int> &labels); // labels: dofmap of the local rows; A is borrowed for level 0
   // row block of a partitioned matrix: columns [owned | ghosts], hA its halo plan, part the row starts of all ranks
   void        setup_dist(const DCsr &A, const HaloPlan &hA, const std::vector<long long> &part, const std::vector<long long> &ghost_gids,
                          const std::vector<int> &labels);
   void        solve(const double *b, double *x, bool zero_guess);   // max_iter cycles
   // preconditioner.reuse: a kept MGR applied to a later system takes level 0 from the matrix of the call (as hypre_MGRSolve does)
   bool        bound_to(const DCsr &A) const { return !lv.empty() && lv[0].A == &A && A.nrows == a0_dims[0] && A.ncols == a0_dims[1] && A.nnz == a0_dims[2]; }
   void        rebind(const DCsr &A, const HaloPlan *hA);
   int         num_reduction_levels() const { return (int)lv.size(); }
   const DCsr &matrix(int level, int which) const; // which 0 operator (level == reduction levels: coarsest), 1 P, 2 R
   size_t      vec_len0() const { return lv.empty() ? 0 : (size_t)std::max(lv[0].A->ncols, lv[0].n); }
   MgrParams   prm;

 private:
   struct Level {
      DCsr            A_own;
      const DCsr     *A = nullptr;
      HaloPlan        hA_own, hP;
      const HaloPlan *hA = nullptr;
      size_t          flen = 0; // length the level's f / u need on behalf of the finer level's P
      DCsr            P, R;
      DArray<int>    labels, cf, cidx;
      DArray<double> dinvF, dinvG, f, u, u2, t, ilu_r, ilu_c;
      GsPlan         gs;
      std::unique_ptr<Ilu> gilu; // g_relaxation ilu
      // f_relaxation amg: A_FF (columns [owned F | ghost F]), its halo plan and partition, the AMG on it
      DCsr                   Aff;
      HaloPlan               hFF;
      std::vector<long long> fpart, fghosts;
      DArray<int>            fidx;
      std::unique_ptr<Amg>   famg;
      std::unique_ptr<Ilu>   filu; // f_relaxation ilu
      DArray<double>         rF, eF;
      int                    nf = 0;
      int            n = 0, nc = 0;
   };
   double *cycle(int l, const double *f, double *u, bool zero);
   std::vector<Level>     lv;
   DCsr                   Ac;
   HaloPlan               hAc, hA0_none; // (hA0_none: the empty plan of a one-rank setup)
   std::vector<long long> cparts, cghosts_;
   size_t                 coarse_len = 0;
   int                    a0_dims[3] = {0, 0, 0};
   std::unique_ptr<Amg>   camg;
   std::unique_ptr<Ilu>   cilu; // coarsest_level ilu
   DArray<double>         fc, uc, cilu_r, cilu_c;
};

// ---- setup kernels (hda_amg_setup.hip); exposed for per-kernel parity tests -------------
// hypre_BoomerAMGCreateS: smask[k] = 1 iff entry k of A is a strong connection.
void amg_strength(const DCsr &A, double theta, double max_row_sum, unsigned char *smask, const int *dof = nullptr); // dof: function of every unknown (systems AMG), device
// hypre_BoomerAMGCoarsenPMIS: cf[i] = 1 C, -1 F, -3 special F. row_offset = global id of row 0.
void amg_pmis(const DCsr &A, const unsigned char *smask, uint64_t seed, int level,
              long long row_offset, int *cf);
// hypre_BoomerAMGCoarsenHMIS (coarsen type 10) on the row blocks part (V + 1 row starts; empty = one block): Ruge first pass per
// block, interior C points kept, PMIS from there
void amg_hmis(const DCsr &A, const unsigned char *smask, const std::vector<int> &part, uint64_t seed, int level, int *cf);
// hypre_BoomerAMGBuildExtPIInterp (interp_type 6 / 17) or hypre_BoomerAMGBuildDirInterp with separation of weights (3),
// then InterpTruncation: P (nrows x nc), rows column-sorted.
void amg_interp_extpi(const DCsr &A, const unsigned char *smask, const int *cf, int pmax, // trailing dof: as amg_strength
                      double trunc_factor, DCsr &P, const int *dof = nullptr, int interp_type = 6);
// aggressive coarsening (hda_amg_agg.hip): second strength graph among the C points of cf (c1: rank of every C point), second PMIS
// pass folded into cf, multipass interpolation
void amg_second_strength(const DCsr &A, const unsigned char *smask, const int *cf, int num_paths, DCsr &S2, DArray<int> &c1);
void amg_coarsen_second_pass(const DCsr &A, const unsigned char *smask, int num_paths, uint64_t seed, int level, int *cf);
void amg_interp_multipass(const DCsr &A, const unsigned char *smask, const int *cf, DCsr &P);
void amg_truncate_rows(DCsr &P, int pmax, double trunc_factor); // hypre_BoomerAMGInterpTruncation on finished, column-sorted rows
// mm-ext+i (interp type 17): the matrix-matrix form of extended+i, W = -D^-1 (I + B) A^s_FC, + InterpTruncation (hda_amg_agg.hip)
void amg_interp_mm_extpi(const DCsr &A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor, DCsr &P,
                         const int *dof = nullptr);
// hypre_ParCSRMatMat-style product C = X*Y, deterministic accumulation order, rows sorted.
void spgemm(const DCsr &X, const DCsr &Y, DCsr &C);
// hypre_BoomerAMGBuildCoarseOperator: Ac = R*(A*P) with R = P^T
void amg_rap(const DCsr &A, const DCsr &P, const DCsr &R, DCsr &Ac);

} // namespace hda
