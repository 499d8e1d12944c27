// Tiled large-system path (placeholder until implemented)
#pragma once
#include "epnn_host.h"
struct PairSource;
static int large_plan(epnn_handle *h) { (void)h; return 0; }
static int launch_large(epnn_handle *h, const PairSource &S);
