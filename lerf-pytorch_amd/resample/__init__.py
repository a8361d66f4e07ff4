"""Mirror of the LUT-evaluation functions of the reference's `resample` package."""
