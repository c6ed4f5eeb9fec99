/* tests/atan2f_restatement.c -- the fdlibm atan2f/atanf operation sequence that k5_fm (kernels.hip: atan2f_ref) executes on
 * the device, as plain C, compared bit for bit with the host libm (the one the reference links against).  Test infrastructure. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
/* fdlibm s_atanf.c */
static const float atanhi[] = {4.6364760399e-01f,7.8539812565e-01f,9.8279368877e-01f,1.5707962513e+00f};
static const float atanlo[] = {5.0121582440e-09f,3.7748947079e-08f,3.4473217170e-08f,7.5497894159e-08f};
static const float aT[] = {3.3333334327e-01f,-2.0000000298e-01f,1.4285714924e-01f,-1.1111110449e-01f,9.0908870101e-02f,-7.6918758452e-02f,6.6610731184e-02f,-5.8335702866e-02f,4.9768779427e-02f,-3.6531571299e-02f,1.6285819933e-02f};
static float my_atanf(float x){
  float w,s1,s2,z; int32_t ix,hx,id;
  hx=(int32_t)f2u(x); ix=hx&0x7fffffff;
  if(ix>=0x4c000000){ if(ix>0x7f800000) return x+x; if(hx>0) return atanhi[3]+atanlo[3]; else return -atanhi[3]-atanlo[3]; }
  if(ix<0x3ee00000){ if(ix<0x31000000){ if(1.0e30f+x>1.0f) return x; } id=-1; }
  else { x=fabsf(x);
    if(ix<0x3f980000){ if(ix<0x3f300000){ id=0; x=(2.0f*x-1.0f)/(2.0f+x);} else { id=1; x=(x-1.0f)/(x+1.0f);} }
    else { if(ix<0x401c0000){ id=2; x=(x-1.5f)/(1.0f+1.5f*x);} else { id=3; x=-1.0f/x; } } }
  z=x*x; w=z*z;
  s1=z*(aT[0]+w*(aT[2]+w*(aT[4]+w*(aT[6]+w*(aT[8]+w*aT[10])))));
  s2=w*(aT[1]+w*(aT[3]+w*(aT[5]+w*(aT[7]+w*aT[9]))));
  if(id<0) return x-x*(s1+s2);
  else { z=atanhi[id]-((x*(s1+s2)-atanlo[id])-x); return (hx<0)?-z:z; }
}
static const float tiny=1.0e-30f,zero=0.0f,pi_o_4=7.8539818525e-01f,pi_o_2=1.5707963705e+00f,pi=3.1415927410e+00f,pi_lo=-8.7422776573e-08f;
static float my_atan2f(float y,float x){
  float z; int32_t k,m,hx,hy,ix,iy;
  hx=(int32_t)f2u(x); ix=hx&0x7fffffff; hy=(int32_t)f2u(y); iy=hy&0x7fffffff;
  if(ix>0x7f800000||iy>0x7f800000) return x+y;
  if(hx==0x3f800000) return my_atanf(y);
  m=((hy>>31)&1)|((hx>>30)&2);
  if(iy==0){ switch(m){case 0:case 1:return y;case 2:return pi+tiny;case 3:return -pi-tiny;} }
  if(ix==0) return (hy<0)?-pi_o_2-tiny:pi_o_2+tiny;
  if(ix==0x7f800000){ if(iy==0x7f800000){ switch(m){case 0:return pi_o_4+tiny;case 1:return -pi_o_4-tiny;case 2:return 3.0f*pi_o_4+tiny;case 3:return -3.0f*pi_o_4-tiny;} } else { switch(m){case 0:return zero;case 1:return -zero;case 2:return pi+tiny;case 3:return -pi-tiny;} } }
  if(iy==0x7f800000) return (hy<0)?-pi_o_2-tiny:pi_o_2+tiny;
  k=(iy-ix)>>23;
  if(k>60) z=pi_o_2+0.5f*pi_lo; else if(hx<0&&k<-60) z=0.0f; else z=my_atanf(fabsf(y/x));
  switch(m){ case 0:return z; case 1:{uint32_t zh=f2u(z); return u2f(zh^0x80000000);} case 2:return pi-(z-pi_lo); default:return (z-pi_lo)-pi; }
}
int main(){
  long bad=0,n=0; srand(1);
  for(long i=0;i<10000000;i++){
    float y=(float)((rand()/(double)RAND_MAX*2-1)*pow(10,(rand()%7)-5));
    float x=(float)((rand()/(double)RAND_MAX*2-1)*pow(10,(rand()%7)-5));
    float a=atan2f(y,x), b=my_atan2f(y,x);
    if(f2u(a)!=f2u(b)){ if(bad<5) printf("%a %a : %a vs %a\n",y,x,a,b); bad++; }
    n++;
  }
  printf("mismatch %ld / %ld\n",bad,n); return 0;
}
