// Stand-in for absl::Mutex (the subset common/task.cc, common/thread_pool.cc and
// constraint_builder_2d.cc use): a mutex whose Await(condition) blocks until the condition
// holds, re-evaluated whenever any lock on the mutex is released.
#ifndef DROPIN_SHIMS_ABSL_MUTEX_H_
#define DROPIN_SHIMS_ABSL_MUTEX_H_
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#define LOCKS_EXCLUDED(...)
#define GUARDED_BY(...)
#define EXCLUSIVE_LOCKS_REQUIRED(...)
#define ABSL_GUARDED_BY(...)
#define ABSL_LOCKS_EXCLUDED(...)
#define ABSL_EXCLUSIVE_LOCKS_REQUIRED(...)
namespace absl {
using Duration = std::chrono::nanoseconds;
template <typename Rep, typename Period>
Duration FromChrono(std::chrono::duration<Rep, Period> d) {
  return std::chrono::duration_cast<Duration>(d);
}
class Condition {
 public:
  // absl::Condition(&callable): a pointer to a callable object returning bool.
  template <typename T, typename = decltype(std::declval<const T&>()())>
  explicit Condition(const T* callable) : eval_([callable] { return (*callable)(); }) {}
  template <typename T>
  Condition(bool (*func)(T*), T* arg) : eval_([func, arg] { return func(arg); }) {}
  template <typename T>
  Condition(T* object, bool (T::*method)()) : eval_([object, method] { return (object->*method)(); }) {}
  template <typename T>
  Condition(const T* object, bool (T::*method)() const)
      : eval_([object, method] { return (object->*method)(); }) {}
  bool Eval() const { return eval_(); }
 private:
  std::function<bool()> eval_;
};
class Mutex {
 public:
  void Lock() { mu_.lock(); }
  void Unlock() { mu_.unlock(); cv_.notify_all(); }
  void Await(const Condition& cond) {
    // called with the mutex held
    std::unique_lock<std::mutex> lock(mu_, std::adopt_lock);
    cv_.wait(lock, [&cond] { return cond.Eval(); });
    lock.release();
  }
  bool AwaitWithTimeout(const Condition& cond, Duration timeout) {
    std::unique_lock<std::mutex> lock(mu_, std::adopt_lock);
    const bool ok = cv_.wait_for(lock, timeout, [&cond] { return cond.Eval(); });
    lock.release();
    return ok;
  }
 private:
  std::mutex mu_;
  std::condition_variable cv_;
};
class MutexLock {
 public:
  explicit MutexLock(Mutex* mu) : mu_(mu) { mu_->Lock(); }
  ~MutexLock() { mu_->Unlock(); }
  MutexLock(const MutexLock&) = delete;
  MutexLock& operator=(const MutexLock&) = delete;
 private:
  Mutex* const mu_;
};
}  // namespace absl
#endif  // DROPIN_SHIMS_ABSL_MUTEX_H_
