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

inline void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  auto rol = [](uint64_t v, int r) { return r ? (v << r) | (v >> (64 - r)) : v; };
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t C[5], D[5], B[25];
    for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
    for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
    // rho + pi: B[y][(2x+3y)%5] = rot(A[x][y]); index = x + 5y
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], ROT[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) st[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y]);
    st[0] ^= RC[rnd];
  }
}

inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t st[25];
  memset(st, 0, sizeof(st));
  size_t off = 0;
  auto absorb = [&](const uint8_t* blk) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w;
      memcpy(&w, blk + 8 * i, 8);  // little-endian host
      st[i] ^= w;
    }
    keccak_f1600(st);
  };
  while (len - off >= rate) {
    absorb(data + off);
    off += rate;
  }
  uint8_t last[136];
  memset(last, 0, sizeof(last));
  memcpy(last, data + off, len - off);
  last[len - off] ^= 0x01;
  last[rate - 1] ^= 0x80;
  absorb(last);
  memcpy(out, st, 32);
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
    std::vector<uint8_t> in;
    in.reserve(64 + buf.size() + 1);
    in.insert(in.end(), state, state + 64);
    in.insert(in.end(), buf.begin(), buf.end());
    in.push_back(0);
    keccak256(in.data(), in.size(), out);
    in.back() = 1;
    keccak256(in.data(), in.size(), out + 32);
    memcpy(state, out, 64);
  }
};

}  // namespace cap
