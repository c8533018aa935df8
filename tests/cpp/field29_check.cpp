// Host-side check of the lazy 29-bit field (cap_amd/csrc/field29.hpp) with bound assertions enabled: reads
// "field op a b" lines (hex) on stdin, prints the raw result; tests/test_field29_host.py compares with Python integers.
#define CAP_FL_CHECK 1
#include "../../cap_amd/csrc/field29.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace cap;
// line: which(q/r) op  a(66 hex digits: up to 261 bits)  b
static fl parse(const char* s){ // hex big-endian string, any length up to 66
  fl r; for(int i=0;i<9;i++) r.v[i]=0;
  int n=strlen(s); // value bits
  // parse into bytes
  unsigned char bytes[40]={0}; int nb=0;
  for(int i=n; i>0; i-=2){ char t[3]={0}; if(i>=2){t[0]=s[i-2];t[1]=s[i-1];} else {t[0]=s[i-1];} bytes[nb++]=strtoul(t,0,16);} 
  for(int bit=0; bit<nb*8; bit++){ if(bytes[bit/8]>>(bit%8)&1){ int limb=bit/29; if(limb>8) limb=8; int off=bit-29*limb; r.v[limb]|=1u<<off; } }
  return r;
}
static void print(const fl& a){ // print as integer hex via limbs (may be unnormalized: accumulate in 320-bit)
  unsigned __int128 acc=0; unsigned char out[48]={0}; int bits=0; int pos=0;
  // simple big accumulate: value = sum v[i]<<29i ; use array of 64-bit
  unsigned long long w[6]={0};
  for(int i=0;i<9;i++){ int bit=29*i; int k=bit/64, off=bit%64; unsigned __int128 t=(unsigned __int128)a.v[i]<<off; 
    unsigned __int128 s=(unsigned __int128)w[k]+(unsigned long long)t; w[k]=(unsigned long long)s; unsigned __int128 c=(s>>64)+(t>>64);
    for(int kk=k+1;kk<6&&c;kk++){ s=(unsigned __int128)w[kk]+(unsigned long long)c; w[kk]=(unsigned long long)s; c=(s>>64)+(c>>64);} }
  for(int k=5;k>=0;k--) printf("%016llx",w[k]); printf("\n");
}
template<class F> void run(char op, fl a, fl b){
  fl r;
  switch(op){
    case 'm': r=F::mul(a,b); break;
    case 'q': r=F::sqr(a); break;
    case 'a': r=F::add(a,b); break;
    case 's': r=F::sub(a,b); break;
    case 'w': r=F::weak_reduce(a); break;
    case 'c': r=F::canonical(a); break;
    case 'z': r=F::zero(); r.v[0]=F::is_zero(a); break;
    case 'p': r=F::unpack(F::pack(a)); break;
    case 'e': r=F::from_ext(F::pack(a)); break;   // a < 2^256
    case 'x': r=F::unpack(F::to_ext(a)); break;
    case 't': r=F::to_mont(F::pack(a)); break;
    case 'f': r=F::unpack(F::from_mont(a)); break;
    case 'M': r=F::mul_add_mul(a,b,b,a); break;
    case 'E': r=F::zero(); r.v[0]=F::eq(a,b); break;
    case 'i': r=F::inv(F::normalize(a)); break;
    // constant-multiplicand product: b = the plain canonical constant w; its pair (w, wq) is derived as the NTT tables do
    case 'S': { fl t=F::canonical(F::to_mont(F::pack(b))); r=F::mul_shoup(a,b,F::shoup_quotient(t)); break; }
    case 'Q': { fl t=F::canonical(F::to_mont(F::pack(b))); r=F::shoup_quotient(t); break; }
    case '8': r=F::sub8p(a,b); break;
  }
  print(r);
}
int main(){ char w,op; char sa[128],sb[128];
  while(scanf(" %c %c %s %s",&w,&op,sa,sb)==4){ fl a=parse(sa), b=parse(sb); if(w=='q') run<Fq29>(op,a,b); else run<Fr29>(op,a,b);} }
