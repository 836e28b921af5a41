// placeholder
#include "gf3rx_demod.h"
