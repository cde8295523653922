// rmd::CudaException, source-compatible with the reference's include/rmd/cuda_exception.cuh:27-45:
// a std::exception carrying a message and a runtime error code.  Here the code is an RMD_HIP_ERR_*
// value of the C ABI and what() owns its string (the reference returns a pointer into a temporary).
#ifndef RMD_CUDA_EXCEPTION_CUH_
#define RMD_CUDA_EXCEPTION_CUH_

#include <exception>
#include <string>

#include <rmd_hip.h>

namespace rmd {

struct CudaException : public std::exception {
  CudaException(const std::string& what, int err) : what_(what), err_(err) {
    text_ = "CudaException: " + what_ + "\n";
    if (err_ != RMD_HIP_OK) {
      const char* detail = rmd_hip_last_error();
      text_ += "rmd_hip error code: " + std::to_string(err_) + " (" + (detail ? detail : "") + ")\n";
    }
  }
  virtual ~CudaException() throw() {}
  virtual const char* what() const throw() { return text_.c_str(); }
  std::string what_;
  int err_;

 private:
  std::string text_;
};

namespace detail {
inline void throw_on_error(int rc, const char* what) {
  if (rc != RMD_HIP_OK) throw CudaException(what, rc);
}
}  // namespace detail

}  // namespace rmd

#endif  // RMD_CUDA_EXCEPTION_CUH_
