/* tests/sincosf_restatement.c -- the operation sequence of glibc 2.35's sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h:
 * double-precision polynomials on the reduced argument; coefficients = __sincosf_table as found in this image's libm.so.6) that
 * kv2_engine (kernels.hip: sincosf_ref) executes on the device for std::polar(1.0f, f * 2 pi) of V2::FreqOffset::Derotate, as plain C,
 * compared bit for bit with the host libm (the one the reference links against).  glibc selects the function by CPU (ifunc): on CPUs
 * with FMA -- the GPU hosts' -- the variant in which every a * b + c is one fused operation; -DUSE_FMA restates that one (what the
 * device does: __fma_rn), without it 5 of 2 x 10^7 results differ in the last bit.  Test infrastructure. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, s1, c2, s2, c3, s3, c4; } sc_t;
static const sc_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, 0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, -0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13, 0x1.99343027bf8c3p-16},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, -0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, 0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13, -0x1.99343027bf8c3p-16}};
static inline uint32_t asuint(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline uint32_t abstop12(float x){return (asuint(x)>>20)&0x7ff;}
#ifdef USE_FMA
#define MA(a,b,c) fma((a),(b),(c))
#else
#define MA(a,b,c) ((a)*(b)+(c))
#endif
static inline float poly(double x,double x2,const sc_t*p,int n){ /* sinf_poly */
  if((n&1)==0){ double x3=x*x2; double s1=MA(x2,p->s3,p->s2); double x7=x3*x2; double s=MA(x3,p->s1,x); return (float)MA(x7,s1,s); }
  else { double x4=x2*x2; double c2=MA(x2,p->c4,p->c3); double c1=MA(x2,p->c1,p->c0); double x6=x4*x2; double c=MA(x4,p->c2,c1); return (float)MA(x6,c2,c); }
}
static inline double reduce_fast(double x,const sc_t*p,int*np){ double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return MA(-(double)n,p->hpi,x); }
static float my_sin_or_cos(float y,int iscos){ /* |y| < 120: what Derotate can ask for is |y| <= pi / 2 */
  double x=y; const sc_t*p=&T[0]; int n;
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ double s=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return iscos?1.0f:y; return poly(x,s,p,iscos); }
  x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if(n&2)p=&T[1]; return poly(x*s,x*x,p,n^iscos);
}
int main(void){ long bad=0,N=10000000; uint64_t st=88172645463325252ull;
  for(long i=0;i<N;i++){ st^=st<<13; st^=st>>7; st^=st<<17; float y=(float)((double)(st>>11)/9007199254740992.0*2.0-1.0)*((i&1)?1.7f:100.0f);
    if(i%7==0) y*=1e-3f; if(i%1001==0) y*=1e-9f;
    if(asuint(my_sin_or_cos(y,0))!=asuint(sinf(y))) bad++;
    if(asuint(my_sin_or_cos(y,1))!=asuint(cosf(y))) bad++; }
  printf("mismatch %ld / %ld\n",bad,2*N); return bad!=0; }
