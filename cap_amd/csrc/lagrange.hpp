// Lagrange-form commit key of a domain of 2^log_n points: see lagrange.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msm.hpp"

namespace cap {

// Builds the window tables of the n + 3 points  [L_0(tau)] G .. [L_(n-1)(tau)] G, [tau^(n+e) - tau^e] G for e = 0, 1, 2
// (n = 2^log_n) from the monomial SRS table `srs` (>= n + 3 points) on `stream`, and waits for them.  The MSM of a
// column's n values followed by its blinders (two for a wire, three for the permutation product) on this table is
// jf-plonk's commitment to the blinded polynomial.
int lagrange_build(const MsmBases& srs, uint32_t log_n, MsmBases* out, hipStream_t stream);

}  // namespace cap
