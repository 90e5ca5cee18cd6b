// hda_krylov.hip -- PCG and GMRES with all vectors and recurrence scalars resident in HBM.
// The host only sees one 8-byte residual norm per iteration, copied asynchronously while
// the next V-cycle is already enqueued (hypre applies the preconditioner BEFORE testing
// convergence too -- SURVEY App. A.1 -- so the iteration count is unchanged).
#include "hda_krylov.h"

#include <cmath>

namespace hda {

#define g_last_precond_calls (RankState<int, 5>::get())
int last_precond_calls() { return g_last_precond_calls; }

// A Krylov solve that runs INSIDE the preconditioner call of another one (MGR's nested components, reference
// src/internal/krylov.c, mgr.c:3938-3960, 4253-4275) shares the context's device scalars and block partials with its caller.
// PCG keeps <r,r> partials (slot 3) and the gamma of the previous iteration alive across its preconditioner call (fused dots),
// so the nested solve saves the caller's scalar block and partial slots on entry and puts them back on exit: two small
// device copies per nested solve, nothing on the outermost one.
namespace {
struct NestedScope {
   static thread_local int depth;
   DArray<double>          save;
   bool                    nested;
   static constexpr size_t kScal = Context::kNumScalars, kPart = (size_t)Context::kNumSlots * kRedBlocks;
   NestedScope() : nested(depth++ > 0)
   {
      if (!nested) return;
      Context &ctx = Context::get();
      save.alloc(kScal + kPart);
      HDA_HIP(hipMemcpyAsync(save.data(), ctx.scalars, kScal * sizeof(double), hipMemcpyDeviceToDevice, ctx.stream));
      HDA_HIP(hipMemcpyAsync(save.data() + kScal, ctx.partials, kPart * sizeof(double), hipMemcpyDeviceToDevice, ctx.stream));
   }
   ~NestedScope()
   {
      depth--;
      if (!nested) return;
      Context &ctx = Context::get();
      (void)hipMemcpyAsync(ctx.scalars, save.data(), kScal * sizeof(double), hipMemcpyDeviceToDevice, ctx.stream);
      (void)hipMemcpyAsync(ctx.partials, save.data() + kScal, kPart * sizeof(double), hipMemcpyDeviceToDevice, ctx.stream);
   }
};
thread_local int NestedScope::depth = 0;
} // namespace

double pcg_iteration_bytes(const DCsr &A, bool format)
{
   const double n = A.nrows;
   // SpMV + fused <s,p> (reads p again: 8n) ; x,r update + <r,r> : 48n ; p = z + beta p : 24n
   return (matrix_stream_bytes(A, format) + rowptr_stream_bytes(A, format) + 8.0 * A.ncols + 8.0 * n) + 8.0 * n + 48.0 * n + 24.0 * n;
}

// Single-reduction PCG (Chronopoulos & Gear 1989), opt-in: HDA_PCG_SINGLE_REDUCE=1.  hypre's recurrence needs three global sums per
// iteration at two points (<s,p> before the update; <r,z> and <r,r> after the preconditioner); here gamma = <r,u>, delta = <w,u>
// (w = A u, u = M^-1 r) and <r,r> are formed together and travel in ONE all-reduce of three doubles (C2 of SURVEY 2.4), alpha comes
// from the recurrence alpha = gamma / (delta - beta gamma / alpha_old).  Price: a second vector recurrence (s = w + beta s, +24 B per
// row) and a different rounding -- iteration counts within 1 of the standard loop, not bit-equal histories; which is why it is off
// until a measurement on several GPUs says the latency of an all-reduce matters.  Two-norm stopping test only.
static KrylovResult pcg_single_reduction(const LinOp &op, const PrecondFn &M, const KrylovParams &kp, const double *b, double *x)
{
   Context     &ctx = Context::get();
   const DCsr  &A   = *op.A;
   const int    n   = A.nrows;
   KrylovResult res;
   const size_t vl = std::max<size_t>(op.veclen, 1);
   DArray<double> r(vl), u(vl), w(vl), p(vl), s(vl);
   constexpr int T0 = S_GMRES, T1 = S_GMRES + 3, S_ALPHA = S_GMRES + 6; // (gamma, <r,r>, delta) triples of alternating iterations
   auto precond = [&](const double *rr, double *zz, int slot) {
      res.precond_calls++;
      if (M) M(rr, zz, slot);
      else
      {
         copy(n, rr, zz);
         if (slot >= 0) dot(n, rr, zz, slot);
      }
   };
   dot(n, b, b, 0);
   finalize(0, S_BB);
   const double bi_prod = read_scalar(S_BB);
   if (bi_prod == 0.0)
   {
      copy(n, b, x);
      ctx.sync();
      res.hist.push_back(0.0);
      return res;
   }
   double eps = kp.rtol * kp.rtol;
   if (kp.atol * kp.atol / bi_prod > eps) eps = kp.atol * kp.atol / bi_prod;
   copy(n, x, p.data()); // (x has no ghost tail: stage it)
   residual(A, p.data(), b, r.data(), op.halo);
   fill((int)vl, 0.0, p.data());
   fill((int)vl, 0.0, s.data());
   dot(n, r.data(), r.data(), 3);
   precond(r.data(), u.data(), 2);
   spmv_dot(A, u.data(), w.data(), u.data(), 4, op.halo);
   finalize_n(2, 3, T0);
   read_scalars_async(T0, 3);
   wait_event(ctx.ev);
   double i_prod = ctx.host_scalars[T0 + 1];
   res.hist.push_back(std::sqrt(std::fabs(i_prod)));
   int it = 0;
   while (it + 1 <= kp.max_iter)
   {
      it++;
      const int tn = (it & 1) ? T0 : T1, tnext = (it & 1) ? T1 : T0; // the triple this step consumes / the one it produces
      const double delta = ctx.host_scalars[tn + 2];
      if (delta == 0.0 || !std::isfinite(delta)) { it--; break; } // breakdown, as hypre's <s,p> == 0
      cg_single_step(n, tn, tnext, S_ALPHA, it == 1, u.data(), w.data(), p.data(), s.data(), x, r.data(), 3);
      // near the tolerance <r,r> is reduced and tested first, so that the V-cycle and the product that would only serve the next step
      // are not run (one extra small all-reduce in the last iteration or two)
      bool near = false;
      if (res.hist.size() >= 2)
      {
         const double h1 = res.hist.back(), h0 = res.hist[res.hist.size() - 2];
         const double rho2 = (h0 > 0.0) ? (h1 / h0) * (h1 / h0) : 1.0;
         near = h1 * h1 * rho2 < 16.0 * eps * bi_prod;
      }
      if (near)
      {
         finalize(3, tnext + 1);
         read_scalars_async(tnext, 3);
         wait_event(ctx.ev);
         i_prod = ctx.host_scalars[tnext + 1];
         if (i_prod / bi_prod < eps)
         {
            res.hist.push_back(std::sqrt(std::fabs(i_prod)));
            res.converged = true;
            break;
         }
      }
      precond(r.data(), u.data(), 2);
      spmv_dot(A, u.data(), w.data(), u.data(), 4, op.halo);
      finalize_n(2, 3, tnext); // <r,u>, <r,r>, <w,u>: one kernel, ONE all-reduce of three doubles
      read_scalars_async(tnext, 3);
      wait_event(ctx.ev);
      i_prod = ctx.host_scalars[tnext + 1];
      res.hist.push_back(std::sqrt(std::fabs(i_prod)));
      if (kp.print_level >= 2)
         printf("%5d    %e    %f    %e\n", it, res.hist.back(),
                res.hist[res.hist.size() - 2] > 0 ? res.hist.back() / res.hist[res.hist.size() - 2] : 0.0, std::sqrt(i_prod / bi_prod));
      if (i_prod / bi_prod < eps) { res.converged = true; break; }
   }
   ctx.sync();
   gs_free_check();
   res.iters     = it;
   res.final_rel = std::sqrt(std::fabs(i_prod) / bi_prod);
   return res;
}

KrylovResult pcg(const LinOp &op, const PrecondFn &M, const KrylovParams &kp, const double *b, double *x)
{
   NestedScope  scope;
   if (kp.two_norm && !scope.nested)
   {
      const char *sr = getenv("HDA_PCG_SINGLE_REDUCE");
      if (sr && atoi(sr) != 0)
      {
         KrylovResult r1 = pcg_single_reduction(op, M, kp, b, x);
         g_last_precond_calls = r1.precond_calls;
         return r1;
      }
   }
   Context     &ctx = Context::get();
   const DCsr  &A   = *op.A;
   const int    n   = A.nrows;
   KrylovResult res;
   const size_t vl = std::max<size_t>(op.veclen, 1);
   DArray<double> r(vl), p(vl), s(vl);

   // first-sweep offer of the preconditioner (hda_amg.h FirstSweepFusion): renewed by every application, used by the next update
   FirstSweepFusion &fs = first_sweep_fusion();
   fs                   = FirstSweepFusion();
   struct FsReset { // also on the way out through an exception: a flag left set would make the next solver's first cycle skip its sweep
      ~FsReset() { first_sweep_fusion() = FirstSweepFusion(); }
   } fs_reset;
   const double *z0_dinv = nullptr;
   double       *z0_dest = nullptr;
   bool          z0_self = false; // the offered destination is the output vector of the next application
   auto precond = [&](const double *rr, double *zz, int slot) {
      res.precond_calls++;
      fs.valid = false;
      if (M) M(rr, zz, slot);
      else
      {
         copy(n, rr, zz);
         if (slot >= 0) dot(n, rr, zz, slot);
      }
      HDA_REQUIRE(!fs.done, "the preconditioner ignored a first sweep it had offered to take from the caller");
      const bool mine = fs.valid && fs.n == n; // (an offer about an operator of another size is not for this loop)
      z0_dinv = mine ? fs.dinv : nullptr;
      z0_dest = mine ? fs.dest : nullptr;
      z0_self = mine && fs.dest == nullptr;
      fs.valid = false;
   };
   // one rank: the finalize launches of <s,p> and of the <r,z>, <r,r> pair ride on the kernels that consume them (same bits)
   // HDA_FUSE_FINALIZE: 0 none, 1 (default) <s,p> inside the update kernel, 2 also the <r,z>, <r,r> pair inside the direction kernel
   // (measured slower: the host's read-back of <r,r> then waits for the direction kernel instead of running beside it, and the
   // device idles while the host catches up -- 256^3: 33.9 vs 34.4 ms, 64^3: 2.02 vs 2.09 ms, profiles/r03_kernel_experiments.md)
   const char *ffe = getenv("HDA_FUSE_FINALIZE"); // (read per solve: the tests switch it inside one process)
   const int   fl  = (Comm::world().size == 1) ? (ffe ? atoi(ffe) : 1) : 0;
   const bool  fin = fl >= 1, fin2 = fl >= 2;

   // bi_prod = <b,b> (two_norm) or <C b, b>
   double bi_prod;
   if (kp.two_norm)
   {
      dot(n, b, b, 0);
      finalize(0, S_BB);
   }
   else
   {
      precond(b, p.data(), 0);
      finalize(0, S_BB);
   }
   bi_prod = read_scalar(S_BB);
   if (bi_prod == 0.0)
   {
      copy(n, b, x); // hypre: zero rhs => x = b = 0
      ctx.sync();
      res.hist.push_back(0.0);
      return res;
   }
   double eps = kp.rtol * kp.rtol;
   {
      const double a2 = kp.atol * kp.atol / bi_prod;
      if (a2 > eps) eps = a2;
   }
   // r = b - A x ; p = C r ; gamma = <r,p>   (x has no ghost tail: stage it through p)
   copy(n, x, p.data());
   residual(A, p.data(), b, r.data(), op.halo);
   precond(r.data(), p.data(), 2);
   dot(n, r.data(), r.data(), 3);
   finalize_n(2, 2, S_GAMMA0); // <r,z> -> S_GAMMA0, <r,r> -> S_RR0: one kernel, one all-reduce of two doubles
   read_scalars_async(S_GAMMA0, 5);
   wait_event(ctx.ev);
   double i_prod = kp.two_norm ? ctx.host_scalars[S_RR0] : ctx.host_scalars[S_GAMMA0];
   res.hist.push_back(std::sqrt(std::fabs(i_prod)));
   int it = 0;
   std::vector<hipEvent_t> evs;
   // HDA_FUSE_DOTS=0: every inner product gets its own all-reduce (the unfused path the fused one is tested against)
   static const bool fuse_dots = !(getenv("HDA_FUSE_DOTS") && atoi(getenv("HDA_FUSE_DOTS")) == 0);
   while (it + 1 <= kp.max_iter)
   {
      it++;
      const int go = (it & 1) ? S_GAMMA0 : S_GAMMA1, gn = (it & 1) ? S_GAMMA1 : S_GAMMA0, rn = gn + 1;
      if (kp.profile_k1)
      {
         hipEvent_t e0, e1;
         HDA_HIP(hipEventCreate(&e0));
         HDA_HIP(hipEventCreate(&e1));
         HDA_HIP(hipEventRecord(e0, ctx.stream));
         spmv_dot(A, p.data(), s.data(), p.data(), 0, op.halo); // row blocks: the ghost refresh of p runs under the product
         HDA_HIP(hipEventRecord(e1, ctx.stream));
         evs.push_back(e0);
         evs.push_back(e1);
      }
      else
         spmv_dot(A, p.data(), s.data(), p.data(), 0, op.halo);
      if (!fin) finalize(0, S_SP);
      double *z0 = z0_dinv ? (z0_self ? s.data() : z0_dest) : nullptr;
      cg_update(n, go, p.data(), s.data(), x, r.data(), 3, fin ? 0 : -1, z0_dinv, z0);
      // The stopping test of the two-norm variant needs only <r,r>.  hypre applies the preconditioner
      // before testing, so its last V-cycle is computed and thrown away; here, once the history says the
      // tolerance is within reach, the host waits for <r,r> BEFORE enqueueing that V-cycle and skips it
      // on convergence (same x, same iteration count, same history).  Far from convergence the
      // V-cycle is enqueued first so the device never waits for the host, and <r,r> rides with <r,z> in
      // one finalize kernel and one two-double all-reduce (row partitions: 2 all-reduces per iteration
      // instead of 3).
      bool stop = false;
      auto test = [&]() {
         wait_event(ctx.ev);
         const double sp = ctx.host_scalars[S_SP];
         if (sp == 0.0 || !std::isfinite(sp))
         {
            it--; // hypre: <s,p> == 0 is a breakdown, the update was not meaningful
            stop = true;
            return;
         }
         i_prod = kp.two_norm ? ctx.host_scalars[rn] : ctx.host_scalars[gn];
         res.hist.push_back(std::sqrt(std::fabs(i_prod)));
         if (kp.print_level >= 2)
            printf("%5d    %e    %f    %e\n", it, res.hist.back(),
                   res.hist[res.hist.size() - 2] > 0 ? res.hist.back() / res.hist[res.hist.size() - 2] : 0.0,
                   std::sqrt(i_prod / bi_prod));
         if (i_prod / bi_prod < eps)
         {
            res.converged = true;
            stop          = true;
         }
      };
      bool near = false;
      if (kp.two_norm && res.hist.size() >= 2)
      { // predicted <r,r> of this iteration from the last reduction factor, with a margin of 16
         const double h1 = res.hist.back(), h0 = res.hist[res.hist.size() - 2];
         const double rho2 = (h0 > 0.0) ? (h1 / h0) * (h1 / h0) : 1.0;
         near = h1 * h1 * rho2 < 16.0 * eps * bi_prod;
      }
      if (near)
      {
         finalize(3, rn);
         read_scalars_async(S_GAMMA0, 5);
         test();
         if (stop) break;
         fs.done = z0 != nullptr;
         fs.in   = r.data();
         fs.out  = s.data();
         precond(r.data(), s.data(), 2);
         if (fin2) cg_direction(n, go, gn, s.data(), p.data(), 2); // (finishes slot 3 into rn once more: the same sum)
         else
         {
            finalize(2, gn);
            cg_direction(n, go, gn, s.data(), p.data());
         }
         continue;
      }
      if (!fuse_dots) finalize(3, rn);
      fs.done = z0 != nullptr;
      fs.in   = r.data();
      fs.out  = s.data();
      precond(r.data(), s.data(), 2);
      if (fin2 && fuse_dots)
      {
         cg_direction(n, go, gn, s.data(), p.data(), 2);
         read_scalars_async(S_GAMMA0, 5);
      }
      else
      {
         if (fuse_dots) finalize_n(2, 2, gn);
         else finalize(2, gn);
         read_scalars_async(S_GAMMA0, 5);
         cg_direction(n, go, gn, s.data(), p.data());
      }
      test();
      if (stop) break;
   }
   fs = FirstSweepFusion();
   ctx.sync();
   gs_free_check();
   for (size_t e = 0; e + 1 < evs.size(); e += 2)
   {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, evs[e], evs[e + 1]) == hipSuccess)
      {
         res.k1_ms_sum += ms;
         res.k1_count++;
      }
      (void)hipEventDestroy(evs[e]);
      (void)hipEventDestroy(evs[e + 1]);
   }
   res.iters     = it;
   res.final_rel = std::sqrt(std::fabs(i_prod) / bi_prod);
   if (!scope.nested) g_last_precond_calls = res.precond_calls;
   return res;
}

static KrylovResult gmres_core(bool flexible, const LinOp &op, const PrecondFn &M, const KrylovParams &kp, const double *b, double *x)
{
   NestedScope  scope;
   Context     &ctx = Context::get();
   const DCsr  &A   = *op.A;
   const int    n = A.nrows, k = std::max(kp.krylov_dim, 1);
   KrylovResult res;
   HDA_REQUIRE(S_GMRES + k + 2 <= Context::kNumScalars, "krylov_dim too large for the scalar block");
   const size_t vlen = std::max<size_t>(op.veclen, 1);
   std::vector<DArray<double>> V((size_t)k + 1);
   for (auto &v : V) v.alloc(vlen);
   DArray<double> w(vlen), r(vlen);
   // FlexGMRES keeps the preconditioned directions z_j = M^-1 v_j and updates x with them
   std::vector<DArray<double>> Z(flexible ? (size_t)k : 0);
   for (auto &z : Z) z.alloc(vlen);
   auto true_residual = [&](double *out) { // out = b - A x, x staged through w for its ghost tail
      copy(n, x, w.data());
      residual(A, w.data(), b, out, op.halo); // row blocks: ghost refresh under the product
   };
   std::vector<double> H((size_t)(k + 1) * k, 0.0), cs((size_t)k), sn((size_t)k), rs((size_t)k + 1);

   auto precond = [&](const double *rr, double *zz) {
      if (M) M(rr, zz, -1);
      else copy(n, rr, zz);
   };
   auto norm2 = [&](const double *v) {
      dot(n, v, v, 0);
      finalize(0, S_TMP);
      return std::sqrt(read_scalar(S_TMP));
   };
   const double b_norm = norm2(b);
   true_residual(V[0].data());
   double r_norm         = norm2(V[0].data());
   const double den_norm = (b_norm > 0.0) ? b_norm : r_norm;
   double       epsilon  = std::max(kp.atol, kp.rtol * den_norm);
   res.hist.push_back(r_norm);
   int iter = 0;
   if (r_norm == 0.0)
   {
      res.converged = true;
      return res;
   }
   while (iter < kp.max_iter)
   {
      rs[0] = r_norm;
      if (r_norm <= epsilon && iter >= kp.min_iter)
      { // (also before the first iteration: hypre_GMRESSolve accepts an initial guess that already meets the tolerance with 0
        // iterations -- asserted by the reference's own tests/test_init_guess.c:170-199, 247-270)
         true_residual(r.data());
         r_norm = norm2(r.data());
         if (r_norm <= epsilon) { res.converged = true; break; }
         copy(n, r.data(), V[0].data());
         rs[0] = r_norm;
      }
      scale(n, 1.0 / r_norm, V[0].data());
      int i = 0;
      while (i < k && iter < kp.max_iter)
      {
         i++;
         iter++;
         double *zi = flexible ? Z[(size_t)i - 1].data() : r.data();
         precond(V[i - 1].data(), zi);
         spmv(A, 1.0, zi, 0.0, nullptr, V[i].data(), op.halo);
         // modified Gram-Schmidt with the coefficients kept on the device
         for (int j = 0; j < i; j++)
         {
            dot(n, V[j].data(), V[i].data(), 0);
            finalize(0, S_GMRES + j);
            axpy_dev(n, S_GMRES + j, -1.0, V[j].data(), V[i].data());
         }
         dot(n, V[i].data(), V[i].data(), 0);
         finalize(0, S_GMRES + i);
         scale_inv_sqrt_dev(n, S_GMRES + i, V[i].data());
         read_scalars_async(S_GMRES, i + 1);
         wait_event(ctx.ev);
         for (int j = 0; j < i; j++) H[(size_t)j * k + (i - 1)] = ctx.host_scalars[S_GMRES + j];
         H[(size_t)i * k + (i - 1)] = std::sqrt(ctx.host_scalars[S_GMRES + i]);
         for (int j = 1; j < i; j++)
         {
            const double hv                  = H[(size_t)(j - 1) * k + (i - 1)];
            H[(size_t)(j - 1) * k + (i - 1)] = cs[j - 1] * hv + sn[j - 1] * H[(size_t)j * k + (i - 1)];
            H[(size_t)j * k + (i - 1)]       = -sn[j - 1] * hv + cs[j - 1] * H[(size_t)j * k + (i - 1)];
         }
         const double hh = H[(size_t)(i - 1) * k + (i - 1)], hn = H[(size_t)i * k + (i - 1)];
         double       gm = std::sqrt(hh * hh + hn * hn);
         if (gm == 0.0) gm = 1.0e-16;
         cs[i - 1] = hh / gm;
         sn[i - 1] = hn / gm;
         rs[i]     = -sn[i - 1] * rs[i - 1];
         rs[i - 1] = cs[i - 1] * rs[i - 1];
         H[(size_t)(i - 1) * k + (i - 1)] = cs[i - 1] * hh + sn[i - 1] * hn;
         r_norm                           = std::fabs(rs[i]);
         res.hist.push_back(r_norm);
         if (kp.print_level >= 2) printf("%5d    %e    %e\n", iter, r_norm, r_norm / den_norm);
         if (r_norm <= epsilon && iter >= kp.min_iter) break;
      }
      rs[i - 1] = rs[i - 1] / H[(size_t)(i - 1) * k + (i - 1)];
      for (int q = i - 2; q >= 0; q--)
      {
         double tt = rs[q];
         for (int j = q + 1; j < i; j++) tt -= H[(size_t)q * k + j] * rs[j];
         rs[q] = tt / H[(size_t)q * k + q];
      }
      if (flexible)
         for (int j = i - 1; j >= 0; j--) axpy(n, rs[j], Z[(size_t)j].data(), x);
      else
      {
         copy(n, V[i - 1].data(), w.data());
         scale(n, rs[i - 1], w.data());
         for (int j = i - 2; j >= 0; j--) axpy(n, rs[j], V[j].data(), w.data());
         precond(w.data(), r.data());
         axpy(n, 1.0, r.data(), x);
      }
      true_residual(V[0].data());
      const double true_norm = norm2(V[0].data());
      if (r_norm <= epsilon)
      {
         r_norm = true_norm;
         if (kp.skip_real_res_check || true_norm <= epsilon) { res.converged = true; break; }
      }
      else
         r_norm = true_norm;
   }
   ctx.sync();
   gs_free_check();
   res.iters     = iter;
   res.final_rel = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
   return res;
}

KrylovResult gmres(const LinOp &op, const PrecondFn &M, const KrylovParams &kp, const double *b, double *x)
{
   return gmres_core(false, op, M, kp, b, x);
}
KrylovResult fgmres(const LinOp &op, const PrecondFn &M, const KrylovParams &kp, const double *b, double *x)
{
   return gmres_core(true, op, M, kp, b, x);
}

// Right-preconditioned BiCGSTAB (van der Vorst 1992) with r0* = r0, stopping on
// ||r|| <= max(atol, rtol ||b||) and accepting only after the true residual has been recomputed.
// The four inner products of an iteration come back to the host (the recurrence needs their quotients
// before the next kernel can be parameterised); everything else stays on the device.
KrylovResult bicgstab(const LinOp &op, const PrecondFn &M, const KrylovParams &kp, const double *b, double *x)
{
   NestedScope  scope;
   const DCsr  &A = *op.A;
   const int    n = A.nrows;
   KrylovResult res;
   const size_t vlen = std::max<size_t>(op.veclen, 1);
   DArray<double> r0(vlen), r(vlen), p(vlen), v(vlen), q(vlen), s(vlen), w(vlen);
   auto true_residual = [&](double *out) {
      copy(n, x, w.data());
      residual(A, w.data(), b, out, op.halo); // row blocks: ghost refresh under the product
   };
   auto precond = [&](const double *rr, double *zz) {
      if (M) M(rr, zz, -1);
      else copy(n, rr, zz);
   };
   auto inner = [&](const double *a, const double *c) {
      dot(n, a, c, 0);
      finalize(0, S_TMP);
      return read_scalar(S_TMP);
   };
   true_residual(r0.data());
   copy(n, r0.data(), r.data());
   copy(n, r0.data(), p.data());
   const double b_norm = std::sqrt(inner(b, b));
   double       rho    = inner(r0.data(), r0.data());
   double       r_norm = std::sqrt(rho);
   const double den    = (b_norm > 0.0) ? b_norm : r_norm;
   const double eps    = std::max(kp.atol, kp.rtol * den);
   res.hist.push_back(r_norm);
   int iter = 0;
   if (r_norm == 0.0)
   {
      res.converged = true;
      return res;
   }
   while (iter < kp.max_iter)
   {
      iter++;
      precond(p.data(), v.data());
      spmv(A, 1.0, v.data(), 0.0, nullptr, q.data(), op.halo);
      const double temp = inner(r0.data(), q.data());
      if (temp == 0.0) break; // breakdown
      const double alpha = rho / temp;
      axpy(n, alpha, v.data(), x);
      axpy(n, -alpha, q.data(), r.data());
      precond(r.data(), v.data());
      spmv(A, 1.0, v.data(), 0.0, nullptr, s.data(), op.halo);
      const double gn = inner(r.data(), s.data()), gd = inner(s.data(), s.data());
      const double gamma = (gn == 0.0 && gd == 0.0) ? 0.0 : gn / gd;
      axpy(n, gamma, v.data(), x);
      axpy(n, -gamma, s.data(), r.data());
      r_norm = std::sqrt(inner(r.data(), r.data()));
      res.hist.push_back(r_norm);
      if (kp.print_level >= 2) printf("%5d    %e    %e\n", iter, r_norm, r_norm / den);
      if (r_norm <= eps && iter >= kp.min_iter)
      {
         true_residual(r.data());
         r_norm = std::sqrt(inner(r.data(), r.data()));
         if (r_norm <= eps) { res.converged = true; break; }
      }
      if (rho == 0.0 || gamma == 0.0) break; // breakdown
      double beta = 1.0 / rho;
      rho         = inner(r0.data(), r.data());
      beta *= rho;
      // p = r + beta (alpha / gamma) (p - gamma q)
      axpy(n, -gamma, q.data(), p.data());
      scale(n, beta * alpha / gamma, p.data());
      axpy(n, 1.0, r.data(), p.data());
   }
   Context::get().sync();
   gs_free_check();
   res.iters     = iter;
   res.final_rel = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
   return res;
}

} // namespace hda
