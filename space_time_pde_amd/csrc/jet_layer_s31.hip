// k_layer instantiations for the stream configuration S1=3, S2=1 (combined second-order stream).
#include "jet_layer_impl.h"
STPDE_DEFINE_LAYER_TU(3, 1)

#if STPDE_STAMP
extern "C" int stpde_stamp_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(g_stamp));
}
#endif
