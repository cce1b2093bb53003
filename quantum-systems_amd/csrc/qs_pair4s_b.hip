// Instantiations of the streamed pair kernel (qs_pair4s.h): complex items, ceil(l/4) = 10 ... 12
#include "qs_pair4s.h"

namespace qs {

int launch_pair4s_b(int n4, const Pair4Args& g, hipStream_t stream) {
    switch (n4) {
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side: one instantiation
        
#else
        case 10: return launch_pair4s<10>(g, stream); case 11: return launch_pair4s<11>(g, stream); case 12: return launch_pair4s<12>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs
