// The C = 192 instances of k_wattn_ring, compiled with MFMA results in the architectural VGPRs (see wattn_ring.hip).
#define RING_TU_192 1
#include "wattn_ring.hip"
