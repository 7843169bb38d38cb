// BLS12-377 build: the endomorphism-accelerated paths (GLV on G1, psi / GLS on G2) are NOT provided for this curve.  The constants below
// exist only so that the shared sources compile; engine.hip forces the plain scalar-multiplication paths (Switches: no_endo) and never
// launches a kernel that reads them.
#pragma once
#define RIPP_ZERO12 {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}
#define RIPP_GLV_BETA RIPP_ZERO12
#define RIPP_GLV_LAMBDA {1u, 0u, 0u, 0u, 1u, 0u, 0u, 0u}
#define RIPP_PSI1_CX0 RIPP_ZERO12
#define RIPP_PSI1_CX1 RIPP_ZERO12
#define RIPP_PSI1_CY0 RIPP_ZERO12
#define RIPP_PSI1_CY1 RIPP_ZERO12
#define RIPP_PSI2_CX0 RIPP_ZERO12
#define RIPP_PSI2_CX1 RIPP_ZERO12
#define RIPP_PSI2_CY0 RIPP_ZERO12
#define RIPP_PSI2_CY1 RIPP_ZERO12
#define RIPP_PSI3_CX0 RIPP_ZERO12
#define RIPP_PSI3_CX1 RIPP_ZERO12
#define RIPP_PSI3_CY0 RIPP_ZERO12
#define RIPP_PSI3_CY1 RIPP_ZERO12
