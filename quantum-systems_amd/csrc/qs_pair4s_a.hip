// Instantiations of the streamed pair kernel (qs_pair4s.h): complex items, ceil(l/4) = 2 ... 9
#include "qs_pair4s.h"

namespace qs {

int launch_pair4s_a(int n4, const Pair4Args& g, hipStream_t stream) {
    switch (n4) {
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side: one instantiation
        case 7: return launch_pair4s<7>(g, stream);
#else
        case 2: return launch_pair4s<2>(g, stream); case 3: return launch_pair4s<3>(g, stream); case 4: return launch_pair4s<4>(g, stream); case 5: return launch_pair4s<5>(g, stream); case 6: return launch_pair4s<6>(g, stream); case 7: return launch_pair4s<7>(g, stream); case 8: return launch_pair4s<8>(g, stream); case 9: return launch_pair4s<9>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs
