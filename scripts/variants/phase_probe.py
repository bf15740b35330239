# mkvar.sh patch: option "phase" for the ideal-overlap probe (scripts/dataflow_probe.py).
#   phase 1: an evaluation enqueues k_setup_items + the broadphase only (counters reset afterwards)
#   phase 2: an evaluation enqueues narrowphase + bristle passes + k_final only, on the candidate list a normal
#            evaluation (phase 0) of the same handle left behind (its count is kept in an unused diagnostic word)
# Results of phase 1 / 2 evaluations are meaningless: this build only answers "how long would the step take if the
# narrowphase of parts of the batch could run beside ONE broadphase launch over the whole batch".
p = 'pfc_hip.hip'; s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a)); s = s.replace(a, b)
rep('''    int opt_poison = 0;''', '''    int opt_phase = 0;
    int opt_poison = 0;''')
rep('''    else if (!std::strcmp(name, "poison")) h->opt_poison = value != 0;''',
    '''    else if (!std::strcmp(name, "poison")) h->opt_poison = value != 0;
    else if (!std::strcmp(name, "phase")) { h->opt_phase = (int)value; h->ghave[0] = h->ghave[1] = false; }''')
# setup + broadphase only when phase != 2
rep('''    hipLaunchKernelGGL(k_setup_items, dim3(grid_for(n_items, 128, 1 << 20)), dim3(128), 0, st, ea);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_SETUP], st));
''', '''    int *saved_cc = (int *)(h->stamps.p + 15);
    if (h->opt_phase != 2) {
    hipLaunchKernelGGL(k_setup_items, dim3(grid_for(n_items, 128, 1 << 20)), dim3(128), 0, st, ea);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_SETUP], st));
''')
rep('''    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_BP], st));

    NpArgs np;''', '''    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_BP], st));
    }   // phase != 2
    if (h->opt_phase == 1) {
        HIP_TRY(h, hipMemsetAsync(h->ctr.p, 0, sizeof(int) * h->ctr.cap, st));
        HIP_TRY(h, hipMemsetAsync(h->status.p, 0, sizeof(unsigned) * 4, st));
        HIP_TRY(h, hipGetLastError());
        return PFC_OK;
    }
    if (h->opt_phase == 0) HIP_TRY(h, hipMemcpyAsync(saved_cc, ccount, sizeof(int), hipMemcpyDeviceToDevice, st));
    if (h->opt_phase == 2) HIP_TRY(h, hipMemcpyAsync(ccount, saved_cc, sizeof(int), hipMemcpyDeviceToDevice, st));

    NpArgs np;''')
open(p, 'w').write(s)
