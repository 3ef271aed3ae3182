PROGRAM cdfzonalmean
#include "cdfzonal.inc"
END PROGRAM cdfzonalmean
