"""hypotremormcmc_amd -- MI355X-native likelihood inner loop of HypoTremorMCMC step 5.

Product code: csrc/ (HIP kernels + C ABI -> lib/libhtm_hip.so) and thin host mirrors of the reference
interfaces (forward, model, chains/parallel, param, obs_data, driver).  Nothing here imports oracle/.
"""
__all__ = ["synth"]
