/* Every public header, in one translation unit, as C99 -pedantic and as C++: they must stand on their own. */
#include "spangpu.h"
#include "spangpu_spandsp.h"
#include "spangpu_prims.h"
#include "spangpu_refstate.h"

int main(void)
{
    return (spangpu_device_count() >= 0  ||  spangpu_last_error() != 0)  ?  0  :  1;
}
