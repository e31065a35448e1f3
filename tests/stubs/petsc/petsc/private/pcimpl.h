/* test stub: see petsc_stub.h */
#include "petsc_stub.h"
