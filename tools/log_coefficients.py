"""Coefficients of htm_log (hypotremormcmc_amd/csrc/htm_device.hpp): near-minimax fit of
P(z) = (log((1+s)/(1-s)) - 2s) / s^3, z = s^2, on [0, ((sqrt2-1)/(sqrt2+1))^2], and the split of ln 2 into a 32-bit high part and
the rest.  python tools/log_coefficients.py  (needs mpmath; nothing else in the repo does)"""
import mpmath as mp
mp.mp.prec = 200
zmax = ((mp.sqrt(2)-1)/(mp.sqrt(2)+1))**2
def P(z):
    if z == 0: return mp.mpf(2)/3
    s = mp.sqrt(z)
    return (mp.log((1+s)/(1-s)) - 2*s)/(s*z)
for deg in (5,6,7):
    c, err = mp.chebyfit(P, [0, zmax*1.0001], deg+1, error=True)
    # chebyfit returns highest degree first
    print(deg, float(err), float(err*zmax/2))
    if deg == 6:
        co = c[::-1]
        for k,v in enumerate(co):
            print(k, mp.nstr(v, 25), float(v).hex())
ln2 = mp.log(2)
import struct
hi = float(ln2)
# ln2_hi: 32 significant bits
b = struct.unpack('<Q', struct.pack('<d', hi))[0] & ~((1<<21)-1)
hi = struct.unpack('<d', struct.pack('<Q', b))[0]
lo = float(ln2 - mp.mpf(hi))
print(hi.hex(), lo.hex(), repr(hi), repr(lo))
