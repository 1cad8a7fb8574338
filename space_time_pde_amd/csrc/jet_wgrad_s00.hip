// k_wgrad instantiations for the stream configuration S1=0, S2=0 (own translation unit).
#include "jet_wgrad_impl.h"
STPDE_DEFINE_WGRAD_TU(0, 0)
