// k_wgrad instantiations for the stream configuration S1=3, S2=4 (own translation unit).
#include "jet_wgrad_impl.h"
STPDE_DEFINE_WGRAD_TU(3, 4)
