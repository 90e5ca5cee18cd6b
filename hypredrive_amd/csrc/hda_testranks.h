// hda_testranks.h -- the in-process test transport (hda_testranks_comm.hip) as hda_thread_ranks.hip sees it
#pragma once

#include "hda_comm.h"

#include <memory>

namespace hda {

std::shared_ptr<void> make_thread_world(int size);
Comm                 *make_thread_comm(int rank, const std::shared_ptr<void> &world);
void                  thread_world_fail(const std::shared_ptr<void> &world);

} // namespace hda
