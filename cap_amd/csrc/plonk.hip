// placeholder until the prover lands
#include "context.hpp"
namespace cap { struct ProvingKey { int dummy; }; }
