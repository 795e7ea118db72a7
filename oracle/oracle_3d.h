// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  3D restatement (filled in below).
#ifndef ORACLE_3D_H_
#define ORACLE_3D_H_
#include "oracle_common.h"
#endif
