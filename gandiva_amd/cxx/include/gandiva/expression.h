#pragma once
#include "gandiva/node.h"
