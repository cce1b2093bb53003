// Instantiations of the streamed fp64 kernel (qs_quad4s.h) for ceil(l/4) = 21 ... 24: two workgroups per item quad.
#include "qs_quad4s.h"

namespace qs {

int launch_quad4s_w2(int n4, const Quad4Args& g, hipStream_t stream) {
    switch (n4) {
#ifndef QS_DEV_FEW_SHAPES      // (development / sanitizer builds of the HOST side: none)
        case 21: return launch_quad4s<21, 2>(g, stream); case 22: return launch_quad4s<22, 2>(g, stream); case 23: return launch_quad4s<23, 2>(g, stream); case 24: return launch_quad4s<24, 2>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs
