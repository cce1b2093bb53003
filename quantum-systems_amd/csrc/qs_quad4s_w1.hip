// Instantiations of the streamed fp64 kernel (qs_quad4s.h) for ceil(l/4) = 17 ... 20: two workgroups per item quad.
#include "qs_quad4s.h"

namespace qs {

int launch_quad4s_w1(int n4, const Quad4Args& g, hipStream_t stream) {
    switch (n4) {
#ifndef QS_DEV_FEW_SHAPES      // (development / sanitizer builds of the HOST side: none)
        case 17: return launch_quad4s<17, 2>(g, stream); case 18: return launch_quad4s<18, 2>(g, stream); case 19: return launch_quad4s<19, 2>(g, stream); case 20: return launch_quad4s<20, 2>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs
