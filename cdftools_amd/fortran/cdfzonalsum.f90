PROGRAM cdfzonalsum
#define ZONAL_SUM 1
#include "cdfzonal.inc"
END PROGRAM cdfzonalsum
