"""MI355X-native keypoint-detection front end (Harris + DoG) behind a C ABI."""
