/* the drop-in header is also valid C (pointers in place of the C++ references) */
#include "librectify_amd.h"
int main(void) { struct LineSegment l = {0, 0, 1, 1, 1, 0, -1}; (void)l; return (int)sizeof(struct ImageTransform) - 80; }
