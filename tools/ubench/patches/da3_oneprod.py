# timing experiment (WRONG RESULTS on purpose): k_dec_a_b3 with one product per step instead of six (every LDS read and DMA piece kept)
PATCH = {'bf16x3.hip': [("""                        const int m = 2 * q + u, pr = m >> 2, mt = (m >> 1) & 1, nt = m & 1;
                        acc[mt][nt] =""", """                        const int m = 2 * q + u, pr = m >> 2, mt = (m >> 1) & 1, nt = m & 1;
                        if (pr < 5) { float qq = af[cb][mt][PA[pr]].x + bf[cb][nt][PB[pr]].y; asm volatile("" :: "v"(qq)); continue; }
                        acc[mt][nt] =""")]}
