/*
 * bsi_oracle.c — CPU ORACLE (test infrastructure, NOT product code): BSI Sum / Range,
 * TopK and GroupBy counting restated from fragment.go, roaring/filter.go, executor.go.
 */
#include "roaring_oracle.h"
