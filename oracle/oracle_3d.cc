// ORACLE — TEST INFRASTRUCTURE ONLY.
#include "oracle_3d.h"
