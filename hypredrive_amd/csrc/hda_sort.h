// hda_sort.h -- workgroup-wide bitonic sort with the keys in registers (NT threads, 256 by default, PER keys each).
// Used by the expand/sort/compress SpGEMM (64-bit keys) and the windowed-CSR plan (32-bit column indices).
#pragma once
#include <hip/hip_runtime.h>

namespace hda {

// compare-exchange of two keys held in registers
template <typename K>
__device__ __forceinline__ void sort_cex(K &a, K &b, const bool up)
{
   const K x = a, y = b;
   const bool               sw = (x > y) == up;
   a = sw ? y : x;
   b = sw ? x : y;
}

// Bitonic sort of NT*PER keys, PER per lane in registers: element i = tid*PER + r lives in k[r] of lane tid.
// Stage (kk, j) compares elements i and i^j, ascending where (i & kk) == 0.  j < PER stays inside a lane's
// registers, j < 64*PER is a lane exchange inside the wave (no LDS traffic, no barrier), and only the stages
// with j >= 64*PER (three of them with four waves) cross waves through xch (NT*PER keys of LDS).
template <int PER, typename K, int NT = 256>
__device__ __forceinline__ void block_sort_regs(K (&k)[PER], K *xch, const int tid)
{
#pragma unroll
   for (int kk = 2; kk <= PER; kk <<= 1)
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1)
#pragma unroll
         for (int r = 0; r < PER; r++)
            if ((r & j) == 0) sort_cex(k[r], k[r | j], ((tid * PER + r) & kk) == 0);
   for (int kk = 2 * PER; kk <= NT * PER; kk <<= 1)
   {
      const bool up = ((tid * PER) & kk) == 0;
      for (int j = kk >> 1; j >= PER; j >>= 1)
      {
         const int  m       = j / PER; // partner lane distance
         const bool keepmin = (((tid & m) == 0) == up);
         if (m >= 64)
         {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PER; r++) xch[r * NT + tid] = k[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PER; r++)
            {
               const K o = xch[r * NT + (tid ^ m)];
               k[r]                       = ((o < k[r]) == keepmin) ? o : k[r];
            }
         }
         else
         {
#pragma unroll
            for (int r = 0; r < PER; r++)
            {
               const K o = __shfl_xor(k[r], m);
               k[r]                       = ((o < k[r]) == keepmin) ? o : k[r];
            }
         }
      }
#pragma unroll
      for (int j = PER >> 1; j > 0; j >>= 1)
#pragma unroll
         for (int r = 0; r < PER; r++)
            if ((r & j) == 0) sort_cex(k[r], k[r | j], up);
   }
}


} // namespace hda
