// Keccak-256 (original padding 0x01) and jf-plonk's SolidityTranscript, host side.
// Replaces `jf_plonk::transcript::SolidityTranscript` (imported at
// src/proof/transfer.rs:39-45; sha3 0.10.1 Keccak256 underneath).  O(1) work per proof;
// stays on the host exactly as in the reference.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace cap {

#define CAP_ROL64(v, r) (((v) << (r)) | ((v) >> (64 - (r))))
// Keccak-f[1600], rounds written out on 25 locals (theta, rho + pi, chi, iota): about half the time of the loop form.
inline void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  uint64_t a00 = st[0], a01 = st[1], a02 = st[2], a03 = st[3], a04 = st[4];
  uint64_t a05 = st[5], a06 = st[6], a07 = st[7], a08 = st[8], a09 = st[9];
  uint64_t a10 = st[10], a11 = st[11], a12 = st[12], a13 = st[13], a14 = st[14];
  uint64_t a15 = st[15], a16 = st[16], a17 = st[17], a18 = st[18], a19 = st[19];
  uint64_t a20 = st[20], a21 = st[21], a22 = st[22], a23 = st[23], a24 = st[24];
  for (int rnd = 0; rnd < 24; rnd++) {
    // theta
    uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
             c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
    uint64_t d0 = c4 ^ CAP_ROL64(c1, 1), d1 = c0 ^ CAP_ROL64(c2, 1), d2 = c1 ^ CAP_ROL64(c3, 1),
             d3 = c2 ^ CAP_ROL64(c4, 1), d4 = c3 ^ CAP_ROL64(c0, 1);
    a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
    a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
    a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
    a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
    a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
    // rho + pi: B[y + 5 ((2x + 3y) mod 5)] = rol(A[x + 5y], ROT[x + 5y])
    uint64_t b00 = a00, b10 = CAP_ROL64(a01, 1), b20 = CAP_ROL64(a02, 62), b05 = CAP_ROL64(a03, 28),
             b15 = CAP_ROL64(a04, 27);
    uint64_t b16 = CAP_ROL64(a05, 36), b01 = CAP_ROL64(a06, 44), b11 = CAP_ROL64(a07, 6), b21 = CAP_ROL64(a08, 55),
             b06 = CAP_ROL64(a09, 20);
    uint64_t b07 = CAP_ROL64(a10, 3), b17 = CAP_ROL64(a11, 10), b02 = CAP_ROL64(a12, 43), b12 = CAP_ROL64(a13, 25),
             b22 = CAP_ROL64(a14, 39);
    uint64_t b23 = CAP_ROL64(a15, 41), b08 = CAP_ROL64(a16, 45), b18 = CAP_ROL64(a17, 15), b03 = CAP_ROL64(a18, 21),
             b13 = CAP_ROL64(a19, 8);
    uint64_t b14 = CAP_ROL64(a20, 18), b24 = CAP_ROL64(a21, 2), b09 = CAP_ROL64(a22, 61), b19 = CAP_ROL64(a23, 56),
             b04 = CAP_ROL64(a24, 14);
    // chi + iota
    a00 = b00 ^ (~b01 & b02) ^ RC[rnd]; a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00);
    a04 = b04 ^ (~b00 & b01);
    a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05);
    a09 = b09 ^ (~b05 & b06);
    a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10);
    a14 = b14 ^ (~b10 & b11);
    a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15);
    a19 = b19 ^ (~b15 & b16);
    a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20);
    a24 = b24 ^ (~b20 & b21);
  }
  st[0] = a00; st[1] = a01; st[2] = a02; st[3] = a03; st[4] = a04; st[5] = a05; st[6] = a06; st[7] = a07;
  st[8] = a08; st[9] = a09; st[10] = a10; st[11] = a11; st[12] = a12; st[13] = a13; st[14] = a14; st[15] = a15;
  st[16] = a16; st[17] = a17; st[18] = a18; st[19] = a19; st[20] = a20; st[21] = a21; st[22] = a22; st[23] = a23;
  st[24] = a24;
}

// incremental sponge (rate 136, original Keccak padding 0x01)
struct Keccak256 {
  uint64_t st[25];
  uint8_t tail[136];
  size_t fill = 0;
  Keccak256() { memset(st, 0, sizeof(st)); }
  void absorb_block(const uint8_t* blk) {
    for (size_t i = 0; i < 17; i++) {
      uint64_t w;
      memcpy(&w, blk + 8 * i, 8);  // little-endian host
      st[i] ^= w;
    }
    keccak_f1600(st);
  }
  void update(const uint8_t* data, size_t len) {
    if (fill) {
      size_t take = 136 - fill < len ? 136 - fill : len;
      memcpy(tail + fill, data, take);
      fill += take;
      data += take;
      len -= take;
      if (fill < 136) return;
      absorb_block(tail);
      fill = 0;
    }
    while (len >= 136) {
      absorb_block(data);
      data += 136;
      len -= 136;
    }
    if (len) {
      memcpy(tail, data, len);
      fill = len;
    }
  }
  void finish(uint8_t out[32]) {  // consumes the object
    memset(tail + fill, 0, 136 - fill);
    tail[fill] ^= 0x01;
    tail[135] ^= 0x80;
    absorb_block(tail);
    memcpy(out, st, 32);
  }
};

inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  Keccak256 k;
  k.update(data, len);
  k.finish(out);
}

struct SolidityTranscript {
  uint8_t state[64];
  std::vector<uint8_t> buf;
  SolidityTranscript() { memset(state, 0, sizeof(state)); }
  void append(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    buf.insert(buf.end(), b, b + n);
  }
  void append_u64_le(uint64_t v) { append(&v, 8); }
  // 64 bytes h0 || h1; the caller reduces the first 48 bytes little-endian mod r
  void challenge_bytes(uint8_t out[64]) {
    // h0 = H(state || buf || 0), h1 = H(state || buf || 1): the two messages differ in their last byte only, so the
    // sponge absorbs state || buf once and is forked for the final byte
    Keccak256 k;
    k.update(state, 64);
    k.update(buf.data(), buf.size());
    Keccak256 k1 = k;
    const uint8_t zero = 0, one = 1;
    k.update(&zero, 1);
    k.finish(out);
    k1.update(&one, 1);
    k1.finish(out + 32);
    memcpy(state, out, 64);
  }
};

}  // namespace cap
