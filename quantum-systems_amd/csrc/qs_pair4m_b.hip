// Instantiations of the streamed pair kernel (qs_pair4s.h): REAL items, ceil(l/4) = 11 ... 14
#include "qs_pair4s.h"

namespace qs {

int launch_pair4m_b(int n4, const Pair4Args& g, hipStream_t stream) {
    switch (n4) {
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side: one instantiation
        
#else
        case 11: return launch_pair4s<11, true>(g, stream); case 12: return launch_pair4s<12, true>(g, stream); case 13: return launch_pair4s<13, true>(g, stream); case 14: return launch_pair4s<14, true>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs
