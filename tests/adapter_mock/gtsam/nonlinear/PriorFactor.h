#include <gtsam/mock_all.h>
