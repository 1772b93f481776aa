# timing experiment (WRONG RESULTS on purpose): k_fc4_b3 with ONE product per step instead of six (all loads kept)
PATCH = {'bf16x3.hip': [
    ("            for (int pr = 0; pr < 6; ++pr)\n", "            for (int pr = 5; pr < 6; ++pr)\n"),
    ("acc[mt][nt], 0, 0, 0);\n            __builtin_amdgcn_sched_barrier(0);", "acc[mt][nt], 0, 0, 0);\n            { float4 q = af[cb][0][1]; q.x += af[cb][0][2].x + af[cb][1][1].y + af[cb][1][2].z + bf[cb][0][1].x + bf[cb][0][2].y + bf[cb][1][1].z + bf[cb][1][2].w; asm volatile(\"\" :: \"v\"(q.x)); }\n            __builtin_amdgcn_sched_barrier(0);"),
]}
