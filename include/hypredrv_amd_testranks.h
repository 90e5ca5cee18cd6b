/* hypredrv_amd_testranks.h -- TEST seam of libhypredrv_amd_testranks.so (hypredrive_amd/csrc/hda_thread_ranks.hip,
 * hda_testranks_comm.hip), a small library on top of libhypredrv_amd.so: the ranks of a row partition as threads of one process.
 * Not part of the drop-in boundary; the product library neither contains nor needs it. */
#ifndef HYPREDRV_AMD_TESTRANKS_H
#define HYPREDRV_AMD_TESTRANKS_H

#ifdef __cplusplus
extern "C" {
#endif

/* `nranks` ranks of a row partition as THREADS of this process -- the library's state is
 * process-global, a thread that joins a thread world gets a private copy until it leaves -- each driving the public HYPREDRV_* sequence of one rank of the reference's
 * examples/src/C_laplacian/laplacian.c:331-468 on the generator's 7-pt Laplacian (global grid n, rank grid P with
 * P[0]*P[1]*P[2] == nranks, `-P 2 2 2` = BASELINE config 3's layout, laplacian.c:561-582).  A GPU box admits six processes
 * on its card, so this is how eight ranks are rehearsed on one GPU.  out16: iterations, converged, final relative residual,
 * |x|_2, |x|_1, |x|_inf, then rank 0's all-reduces / halo exchanges / overlapped exchanges / doubles all-reduced / doubles
 * exchanged of the last solve, V-cycles, partitioned levels, largest difference of the ranks' iteration counts, ranks, 0.
 * x_global (may be NULL): the solution in the generator's block numbering.  Returns 0, or 2 with the ranks' messages in errbuf. */
int hda_thread_ranks_lap7(int nranks, const int n[3], const int P[3], const char *yaml, int nsolves, double out16[16], double *x_global,
                          char *errbuf, int errlen);
/* The same seam for a caller that brings its own threads (tests: Python threads, one per rank, each handing over its row block of an
 * arbitrary CSR matrix through the public API): create a world of nranks, let every thread join as its rank BEFORE its first
 * HYPREDRV_* call, leave when done (failed != 0 releases ranks blocked in a collective with an error), destroy after all have left. */
void *hda_thread_world_create(int nranks);
int hda_thread_world_join(void *world, int rank);
int hda_thread_world_leave(void *world, int failed);
void hda_thread_world_destroy(void *world);
/* Self tests of the seam (returns 0 when the behaviour is as described; the ranks' messages in errbuf).
 * what 0: two thread ranks enter different collectives -- both must return with an error naming the disagreement (not read each other's
 *         stale pointers).  what 2: the same collective with send / receive byte counts that disagree -- an error naming both.  what 1: the device allocator serves a request larger than the driver's free memory by returning the cached
 *         blocks of ANOTHER rank thread (which has cached cache_gb GiB) to the driver. */
int hda_testranks_selftest(int what, double cache_gb, char *errbuf, int errlen);

#ifdef __cplusplus
}
#endif

#endif
