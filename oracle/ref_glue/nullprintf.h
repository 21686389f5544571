/* nullprintf.h -- TEST INFRASTRUCTURE ONLY.  Force-included (after <stdio.h>) when oracle/Makefile compiles the
   reference's echo.c: this snapshot of echo.c printf()s debug text on every sample (echo.c:453,653), which serialises
   threads on the stdio lock and would make the CPU baseline meaningless.  The source file itself is compiled as it is. */
#if !defined(ORACLE_NULLPRINTF_H)
#define ORACLE_NULLPRINTF_H
#define printf(...) ((void) 0)
#endif
