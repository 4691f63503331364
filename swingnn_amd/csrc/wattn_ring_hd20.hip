// The head_dim-20 instances of k_wattn_ring (C = 60 / 120 / 240 padded to 64 / 128 / 256 channels, 32 slots per head).
#define RING_TU_HD20
#include "wattn_ring.hip"
