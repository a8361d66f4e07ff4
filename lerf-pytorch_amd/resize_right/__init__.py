"""Mirror of the reference's `resize_right` package for the LeRF classes
(same class names and call signatures; the work runs in liblerf_hip.so)."""
