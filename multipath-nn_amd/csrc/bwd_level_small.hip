// The dependency-level kernels (bwd_level_k.h) for levels with a block-0 member: that member's image-chunk weight
// gradients run the swapped-role body (bwd_bodies.h, SMALLC).  A translation unit of its own for the build's sake.
#include "bwd_level_k.h"

LevelKern mpnn_level_kernel_smallc(int gkmask) {
    switch (gkmask) {
        case 1: return bwd_level_k<1, 1, true>;
        case 2: return bwd_level_k<2, 1, true>;
        case 3: return bwd_level_k<3, 1, true>;
        case 4: return bwd_level_k<4, 1, true>;
        case 5: return bwd_level_k<5, 1, true>;
        case 6: return bwd_level_k<6, 1, true>;
        case 7: return bwd_level_k<7, 1, true>;
    }
    return nullptr;
}

int mpnn_trace_install_level_small(void *buf) { return mpnn_trace_install(buf); }
