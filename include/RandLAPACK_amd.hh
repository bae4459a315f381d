// Umbrella header of the MI355X-native sketch-and-factor path (mirrors the reference's RandLAPACK.hh for the
// components in scope).
#pragma once
#include "RandLAPACK_amd/rl_exceptions.hh"
#include "RandLAPACK_amd/rl_blaspp.hh"
#include "RandLAPACK_amd/rl_lapackpp.hh"
#include "RandLAPACK_amd/rl_randblas.hh"
#include "RandLAPACK_amd/rl_util.hh"
#include "RandLAPACK_amd/rl_orth.hh"
#include "RandLAPACK_amd/rl_rs.hh"
#include "RandLAPACK_amd/rl_rf.hh"
#include "RandLAPACK_amd/rl_qb.hh"
#include "RandLAPACK_amd/rl_rsvd.hh"
#include "RandLAPACK_amd/rl_cqrrpt.hh"
#include "RandLAPACK_amd/rl_bqrrp.hh"
#include "RandLAPACK_amd/rl_hqrrp.hh"
#include "RandLAPACK_amd/rl_cqrrt.hh"
#include "RandLAPACK_amd/rl_linops.hh"
#include "RandLAPACK_amd/rl_abrik.hh"
