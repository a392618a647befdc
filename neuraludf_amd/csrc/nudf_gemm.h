#pragma once
#include "../../include/nudf.h"
