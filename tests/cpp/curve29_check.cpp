// Host-side check of cap_amd/csrc/curve29.hpp (lazy-field XYZZ formulas, bound assertions on) against the
// saturated 32-bit implementation in curve.hpp: mixed adds with signs, P+P, P-P, infinity, full adds, doublings.
#define CAP_FL_CHECK 1
#include "../../cap_amd/csrc/curve29.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace cap;
static bool same_affine(const g1_affine& a32 /*R-form*/, const g1a& a29){
  fe x32 = Fq::from_mont(a32.x), y32 = Fq::from_mont(a32.y);
  bool inf32 = G1::is_inf(a32), inf29 = G1L::is_inf(a29);
  if (inf32 || inf29) return inf32 == inf29;
  fe x29 = Fq29::from_mont(a29.x), y29 = Fq29::from_mont(a29.y);
  return Fq::eq(x32,x29) && Fq::eq(y32,y29);
}
static g1a conv(const g1_affine& p){ g1a r; if (G1::is_inf(p)) { r.x=Fq29::zero(); r.y=Fq29::zero(); return r;} r.x = Fq29::canonical(Fq29::from_ext(p.x)); r.y = Fq29::canonical(Fq29::from_ext(p.y)); return r; }
int main(){
  // points k*G (32-bit reference)
  const int N = 200;
  std::vector<g1_affine> pts(N);
  g1_affine g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
  g1_xyzz acc = G1::from_affine(g);
  unsigned long long s = 12345;
  for (int i=0;i<N;i++){ pts[i]=G1::to_affine(acc); int reps = 1 + (int)(s % 15); s = (s * 1103515245ULL + 12345ULL) % 2147483648ULL; for(int k=0;k<reps;k++) acc = G1::add_mixed(acc, g); acc = G1::dbl(acc); }
  pts[17] = pts[16];                 // duplicate
  memset(&pts[33], 0, sizeof(g1_affine)); // infinity
  int bad=0;
  // 1. conversions round trip
  for (int i=0;i<N;i++){ g1a a = conv(pts[i]); if(!same_affine(pts[i], a)) { bad++; printf("conv mismatch %d\n", i);} 
     g1_affine m = G1L::store_affine(a); g1a b = G1L::load(m); if(!same_affine(pts[i], b)) {bad++; printf("store/load mismatch %d\n",i);} }
  // 2. running accumulation with signs, incl. doubling (16,17), cancellation, infinity
  g1_xyzz A = G1::inf(); g1x B = G1L::inf();
  for (int i=0;i<N;i++){
    bool neg = (i%5)==3;
    g1_affine p = pts[i]; if (neg) p.y = Fq::neg(p.y);
    A = G1::add_mixed(A, p);
    B = G1L::add_mixed(B, conv(pts[i]), neg);
    if (i==20){ // force acc == q : add the current accumulated point itself (as affine)
      g1_affine cur = G1::to_affine(A); A = G1::add_mixed(A, cur); B = G1L::add_mixed(B, conv(cur)); 
      cur = G1::to_affine(A); g1_affine ncur = cur; ncur.y = Fq::neg(cur.y); A = G1::add_mixed(A, ncur); B = G1L::add_mixed(B, conv(cur), true); // -> infinity
      if (!G1::is_inf(A) || !G1L::is_inf(B)) { bad++; printf("cancellation failed\n"); }
    }
    if(!same_affine(G1::to_affine(A), G1L::to_affine(B))) { bad++; if (bad<5) printf("acc mismatch at %d\n", i); }
    // memory round trip of the accumulator
    B = G1L::load(G1L::store(B));
  }
  // 3. full adds and doublings
  std::vector<g1_xyzz> X(20); std::vector<g1x> Y(20);
  for (int i=0;i<20;i++){ X[i]=G1::inf(); Y[i]=G1L::inf(); for(int k=0;k<5;k++){ X[i]=G1::add_mixed(X[i], pts[i*7+k]); Y[i]=G1L::add_mixed(Y[i], conv(pts[i*7+k])); } }
  for (int i=0;i+1<20;i++){
    g1_xyzz r32 = G1::add(X[i], X[i+1]); g1x r29 = G1L::add(Y[i], Y[i+1]);
    if(!same_affine(G1::to_affine(r32), G1L::to_affine(r29))) { bad++; printf("add mismatch %d\n", i);} 
    r32 = G1::add(X[i], X[i]); r29 = G1L::add(Y[i], Y[i]);     // doubling through add
    if(!same_affine(G1::to_affine(r32), G1L::to_affine(r29))) { bad++; printf("add(P,P) mismatch %d\n", i);} 
    g1_xyzz d32 = X[i]; g1x d29 = Y[i]; for(int k=0;k<13;k++){ d32=G1::dbl(d32); d29=G1L::dbl(d29);} 
    if(!same_affine(G1::to_affine(d32), G1L::to_affine(d29))) { bad++; printf("dbl mismatch %d\n", i);} 
    // jac ext output equals 32-bit to_jac normalised
    g1_jac j29 = G1L::to_jac_ext(d29); g1_affine a_from_j = G1::jac_to_affine(j29);
    g1_affine a32 = G1::to_affine(d32);
    if(!(Fq::eq(a_from_j.x,a32.x)&&Fq::eq(a_from_j.y,a32.y))) { bad++; printf("jac ext mismatch %d\n", i);} 
  }
  // inf + X, X + inf, neg of P
  { g1x r = G1L::add(G1L::inf(), Y[3]); if(!same_affine(G1::to_affine(X[3]), G1L::to_affine(r))) {bad++; printf("inf+X\n");}
    r = G1L::add(Y[3], G1L::inf()); if(!same_affine(G1::to_affine(X[3]), G1L::to_affine(r))) {bad++; printf("X+inf\n");} }
  printf("bad=%d\n", bad);
  return bad!=0;
}
