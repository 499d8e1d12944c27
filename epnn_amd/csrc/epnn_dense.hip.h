#pragma once
#include "epnn_host.h"
