"""Host-side mirror of the reference service interface (`src/upscale`): worker runtime, queue entry, the HIP-backed upscaler service and the model factories."""
