// Instantiations of the streamed pair kernel (qs_pair4s.h): complex items, ceil(l/4) = 13, 14
#include "qs_pair4s.h"

namespace qs {

int launch_pair4s_c(int n4, const Pair4Args& g, hipStream_t stream) {
    switch (n4) {
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side: one instantiation
        
#else
        case 13: return launch_pair4s<13>(g, stream); case 14: return launch_pair4s<14>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs
