// tools/check_div_small.c -- exhaustive check that Markstein's 1 mul + 2 fma sequence (st_div in csrc/pdq_stream.hip, div_small in pdq_fused512.hip) equals the IEEE
// quotient N / d for d = 1..8 and every normal f32 |N| < 4096 (both signs).  Result: exact everywhere except d = 6 below 2^-125.
// gcc -O2 -fopenmp -mfma -ffp-contract=off -o /tmp/check_div_small tools/check_div_small.c -lm   (3 minutes on 8 cores)
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
static inline float div_small(float N, float d, float dinv){ float q0 = N*dinv; float r = fmaf(-q0, d, N); return fmaf(r, dinv, q0);} 
int main(){
  for (int di = 1; di <= 8; di++) {
    const float d = (float)di, dinv = 1.0f/d;
    long bad = 0; uint32_t firstbad = 0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for (int64_t b = 0x00800000; b < 0x45800000; b++) {
      uint32_t u = (uint32_t)b; float N; memcpy(&N,&u,4);
      float want = N / d, got = div_small(N,d,dinv);
      if (memcmp(&want,&got,4)) { bad++; }
      float Nn = -N; want = Nn / d; got = div_small(Nn,d,dinv);
      if (memcmp(&want,&got,4)) { bad++; }
    }
    printf("d=%d bad=%ld\n", di, bad);
  }
  // find the smallest failing exponent per d
  for (int di = 1; di <= 8; di++) {
    const float d = (float)di, dinv = 1.0f/d; uint32_t last = 0;
    for (int64_t b = 0x00800000; b < 0x45800000; b++) { uint32_t u=(uint32_t)b; float N; memcpy(&N,&u,4); float want=N/d, got=div_small(N,d,dinv); if (memcmp(&want,&got,4)) last=u; }
    printf("d=%d largest failing bits=%08x\n", di, last);
  }
}
