"""Physical constants: the RH-code values of /root/reference/constants.py:1-27,
reproduced bit for bit (the hot path's arithmetic depends on them)."""
CLight = 2.99792458E+08
HPlanck = 6.6260755E-34
HC = HPlanck * CLight
KBoltzmann = 1.380658E-23
Amu = 1.6605402E-27
MElectron = 9.1093897E-31
QElectron = 1.60217733E-19
Epsilon0 = 8.854187817E-12
NM_TO_M = 1.0E-09
CM_TO_M = 1.0E-02
KM_TO_M = 1.0E+03
